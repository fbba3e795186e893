// Top-down BFS written against the operator API exactly as a VGL user would (same call sequence as the reference's
// BFS::vgl_top_down, algorithms/bfs/bfs.hpp:6-90): compute(init) -> [scatter(edge_op) -> generate_new_frontier(filter)]*.
#pragma once
#define UNVISITED_VERTEX -1
#define FIRST_LEVEL_VERTEX 1

struct BFS {
    template <typename _T>
    static double vgl_top_down(VGL_Graph &graph, VerticesArray<_T> &levels, int source_vertex)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph);
        VGL_FRONTIER frontier(graph);
        graph_API.change_traversal_direction(SCATTER, levels, frontier);
        Timer tm;
        tm.start();
        auto init_levels = [levels, source_vertex] __VGL_COMPUTE_ARGS__ {
            levels[src_id] = (src_id == source_vertex) ? FIRST_LEVEL_VERTEX : UNVISITED_VERTEX;
        };
        frontier.set_all_active();
        graph_API.compute(graph, frontier, init_levels);
        frontier.clear();
        frontier.add_vertex(source_vertex);
        int current_level = FIRST_LEVEL_VERTEX;
        while (frontier.size() > 0) {
            auto edge_op = [levels, current_level] __VGL_SCATTER_ARGS__ {
                if (levels[src_id] == current_level && levels[dst_id] == UNVISITED_VERTEX) levels[dst_id] = current_level + 1;
            };
            graph_API.scatter(graph, frontier, edge_op);
            auto on_next_level = [levels, current_level] __VGL_GNF_ARGS__ {
                return levels[src_id] == current_level + 1 ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG;
            };
            graph_API.generate_new_frontier(graph, frontier, on_next_level);
            current_level++;
        }
        tm.end();
        performance_stats.print_algorithm_performance_stats("BFS Top-down (operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    // fused fast path of libvgl_hip.so (same result): top-down or direction-optimising
    static double hip_fused(VGL_Graph &graph, VerticesArray<int> &levels, int source_vertex, bool direction_optimising)
    {
        Timer tm;
        tm.start();
        vgl_hip_bfs_stats st;
        VGL_HIP_CALL(vgl_hip_bfs_run(VGL_RUNTIME::ctx(), graph.get_handle(), source_vertex,
                                     direction_optimising ? VGL_HIP_BFS_DIRECTION_OPT : VGL_HIP_BFS_TOP_DOWN, levels.get_ptr(), &st));
        tm.end();
        performance_stats.print_algorithm_performance_stats(direction_optimising ? "BFS direction-optimising (fused)" : "BFS top-down (fused)",
                                                            tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
