// Shiloach-Vishkin against the operator API (call sequence of CC::vgl_shiloach_vishkin, algorithms/cc/shiloach_vishkin.hpp:7-88,
// flags in host-visible memory like gpu_shiloach_vishkin.hpp:30-68).
#pragma once

struct ConnectedComponents {
    template <typename _T>
    static double vgl_shiloach_vishkin(VGL_Graph &graph, VerticesArray<_T> &components)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph);
        VGL_FRONTIER frontier(graph);
        graph_API.change_traversal_direction(SCATTER, components, frontier);
        Timer tm;
        tm.start();
        frontier.set_all_active();
        auto init_components_op = [components] __VGL_COMPUTE_ARGS__ { components[src_id] = src_id; };
        graph_API.compute(graph, frontier, init_components_op);
        int *flags;
        MemoryAPI::allocate_array(&flags, 2);
        int *hook_changes = flags, *jump_changes = flags + 1;
        do {
            hook_changes[0] = 0;
            auto edge_op = [components, hook_changes] __VGL_SCATTER_ARGS__ {
                const int src_val = components[src_id];
                if (src_val < components[dst_id]) { atomicMin(&components[dst_id], src_val); hook_changes[0] = 1; }
            };
            graph_API.scatter(graph, frontier, edge_op);
            do {
                jump_changes[0] = 0;
                auto jump_op = [components, jump_changes] __VGL_COMPUTE_ARGS__ {
                    const int src_val = components[src_id];
                    const int src_src_val = components[src_val];
                    if (src_val != src_src_val) { components[src_id] = src_src_val; jump_changes[0] = 1; }
                };
                graph_API.compute(graph, frontier, jump_op);
            } while (jump_changes[0]);
        } while (hook_changes[0]);
        MemoryAPI::free_array(flags);
        tm.end();
        performance_stats.print_algorithm_performance_stats("CC (Shiloach-Vishkin, operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    // symmetric: the caller knows that every edge is stored in both directions (graphs generated as UNDIRECTED_GRAPH); the labels
    // are the same, computed by the union-find path instead of repeated sweeps
    static double hip_fused(VGL_Graph &graph, VerticesArray<int> &components, bool symmetric = false)
    {
        Timer tm;
        tm.start();
        vgl_hip_cc_stats st;
        if (symmetric) VGL_HIP_CALL(vgl_hip_cc_run_symmetric(VGL_RUNTIME::ctx(), graph.get_handle(), components.get_ptr(), &st));
        else VGL_HIP_CALL(vgl_hip_cc_run(VGL_RUNTIME::ctx(), graph.get_handle(), components.get_ptr(), &st));
        tm.end();
        performance_stats.print_algorithm_performance_stats(symmetric ? "CC (fused, union-find)" : "CC (fused)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
#define CC ConnectedComponents
