// Transitive closure queries ("is b reachable from a?") for a list of vertex pairs -- the two algorithms of the reference
// (algorithms/tc/tc.hpp:5-193) on this backend's operator API:
//   vgl_purdoms   : SCC labels answer the pairs inside one component; the rest is answered on the condensation (one vertex per
//                   component, the edges between different components), one BFS per distinct source component.  Exercises an
//                   EdgesArray written through global_edge_pos in scatter, ParallelPrimitives::copy_if_indexes, reduce(REDUCE_MAX)
//                   and a second VGL_Graph imported from device-resident edges.
//   vgl_bfs_based : one BFS per distinct source on the graph itself (the checker).
// Pairs and answers use the graph's STORED numbering (callers convert with VGL_Graph::reorder).
#pragma once
#include "bfs.hpp"
#include "scc.hpp"
#include <map>
#include <utility>

struct TransitiveClosure {
    using Pairs = std::vector<std::pair<int, int>>;

    // answers for all pairs (position list `which`) that share a source: one traversal, levels read back once
    static void answer_from_source(VGL_Graph &g, VerticesArray<int> &levels, int source, const std::vector<std::pair<int, int>> &targets,
                                   std::vector<int> &answer)
    {
        vgl_hip_bfs_stats st;
        VGL_HIP_CALL(vgl_hip_bfs_run(VGL_RUNTIME::ctx(), g.get_handle(), source, VGL_HIP_BFS_TOP_DOWN, levels.get_ptr(), &st));
        const std::vector<int> h = levels.to_host();
        for (const auto &t : targets) answer[(size_t)t.second] = h[(size_t)t.first] != UNVISITED_VERTEX;
    }

    static double vgl_purdoms(VGL_Graph &graph, const Pairs &vertex_pairs, std::vector<int> &answer)
    {
        Timer tm;
        tm.start();
        VerticesArray<int> components(graph);
        VGL_HIP_CALL(vgl_hip_scc_run(VGL_RUNTIME::ctx(), graph.get_handle(), components.get_ptr(), nullptr));
        const std::vector<int> comp = components.to_host();

        std::map<int, std::vector<std::pair<int, int>>> remaining;        // source component -> (target component, pair index)
        for (size_t i = 0; i < vertex_pairs.size(); i++) {
            const int a = comp[(size_t)vertex_pairs[i].first], b = comp[(size_t)vertex_pairs[i].second];
            if (a == b) answer[i] = 1;
            else remaining[a].push_back({b, (int)i});
        }
        if (!remaining.empty()) {
            VGL_GRAPH_ABSTRACTIONS api(graph);
            VGL_FRONTIER front(graph);
            api.change_traversal_direction(SCATTER, front, components);
            front.set_all_active();
            const long long E = graph.get_edges_count();
            EdgesArray<int> new_src_ids(graph), new_dst_ids(graph);
            auto label_edges = [components, new_src_ids, new_dst_ids] __VGL_SCATTER_ARGS__ {
                const int a = components[src_id], b = components[dst_id];
                new_src_ids[global_edge_pos] = a != b ? a : -1;
                new_dst_ids[global_edge_pos] = a != b ? b : -1;
            };
            api.scatter(graph, front, label_edges);
            new_src_ids.finalize_advance();
            new_dst_ids.finalize_advance();

            long long *edge_indexes = nullptr;
            MemoryAPI::allocate_device_array(&edge_indexes, (size_t)std::max<long long>(E, 1));
            auto between_components = [new_src_ids] __VGL_COPY_IF_INDEXES_ARGS__ { return new_src_ids[idx] != -1; };
            const long long new_edges_count = ParallelPrimitives::copy_if_indexes(between_components, edge_indexes, E);
            auto component_id = [components] __VGL_REDUCE_INT_ARGS__ { return components[src_id]; };
            const int new_vertices_count = api.reduce<int>(graph, front, component_id, REDUCE_MAX) + 1;    // ids are 0 .. max
            std::cout << "condensation: " << new_vertices_count << " vertex ids, " << new_edges_count << " edges" << std::endl;

            if (new_edges_count > 0) {
                EdgesContainer condensed;
                condensed.resize(new_vertices_count, new_edges_count);
                VGL_HIP_CALL(vgl_hip_gather_u32(VGL_RUNTIME::ctx(), new_edges_count, (const int64_t *)edge_indexes, new_src_ids.get_ptr(), condensed.get_src_ids()));
                VGL_HIP_CALL(vgl_hip_gather_u32(VGL_RUNTIME::ctx(), new_edges_count, (const int64_t *)edge_indexes, new_dst_ids.get_ptr(), condensed.get_dst_ids()));
                VGL_Graph ir_graph(CSR_GRAPH);
                ir_graph.import(condensed);
                VerticesArray<int> ir_levels(ir_graph);
                for (const auto &group : remaining) answer_from_source(ir_graph, ir_levels, group.first, group.second, answer);
            } else {
                for (const auto &group : remaining) for (const auto &t : group.second) answer[(size_t)t.second] = 0;
            }
            MemoryAPI::free_device_array(edge_indexes);
        }
        tm.end();
        performance_stats.print_algorithm_performance_stats("TC (Purdom: SCC + condensation BFS)", tm.get_time() / vertex_pairs.size(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time() / vertex_pairs.size(), graph.get_edges_count());
    }

    static double vgl_bfs_based(VGL_Graph &graph, const Pairs &vertex_pairs, std::vector<int> &answer)
    {
        Timer tm;
        tm.start();
        std::map<int, std::vector<std::pair<int, int>>> by_source;
        for (size_t i = 0; i < vertex_pairs.size(); i++) by_source[vertex_pairs[i].first].push_back({vertex_pairs[i].second, (int)i});
        VerticesArray<int> levels(graph);
        for (const auto &group : by_source) answer_from_source(graph, levels, group.first, group.second, answer);
        tm.end();
        performance_stats.print_algorithm_performance_stats("TC (BFS per source)", tm.get_time() / vertex_pairs.size(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time() / vertex_pairs.size(), graph.get_edges_count());
    }
};
#define TC TransitiveClosure
