// Top-down BFS written against the operator API exactly as a VGL user would (same call sequence as the reference's
// BFS::vgl_top_down, algorithms/bfs/bfs.hpp:6-90): compute(init) -> [scatter(discover) -> generate_new_frontier(filter)]*.
#pragma once
#define UNVISITED_VERTEX -1
#define FIRST_LEVEL_VERTEX 1

struct BFS {
    template <typename _T>
    static double vgl_top_down(VGL_Graph &graph, VerticesArray<_T> &levels, int source_vertex)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER front(graph);
        api.change_traversal_direction(SCATTER, levels, front);
        Timer tm;
        tm.start();
        auto mark_source = [levels, source_vertex] __VGL_COMPUTE_ARGS__ {
            levels[src_id] = (src_id == source_vertex) ? FIRST_LEVEL_VERTEX : UNVISITED_VERTEX;
        };
        front.set_all_active();
        api.compute(graph, front, mark_source);
        front.clear();
        front.add_vertex(source_vertex);
        int cur = FIRST_LEVEL_VERTEX;
        while (front.size() > 0) {
            auto discover = [levels, cur] __VGL_SCATTER_ARGS__ {
                if (levels[src_id] == cur && levels[dst_id] == UNVISITED_VERTEX) levels[dst_id] = cur + 1;
            };
            api.scatter(graph, front, discover);
            auto just_found = [levels, cur] __VGL_GNF_ARGS__ {
                return levels[src_id] == cur + 1 ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG;
            };
            api.generate_new_frontier(graph, front, just_found);
            cur++;
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
