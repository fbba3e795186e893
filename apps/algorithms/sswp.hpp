// Single-source widest paths against the operator API (call sequence of SSWP::vgl_dijkstra,
// algorithms/sswp/widest_paths.hpp:5-76: init, then { scatter(edge_op_push); changes } until nothing changes; the `changes`
// word lives in host-visible memory like the reference's GPU variants).  The update uses an integer atomic-max on the f32
// bits (widths are non-negative) so that no update is lost; the fixed point is the same bit pattern either way.
#pragma once

struct WidestPaths {
    template <typename _T>
    static double vgl_dijkstra(VGL_Graph &graph, EdgesArray<_T> &edges_capacities, VerticesArray<_T> &widths, int source_vertex)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph, SCATTER);
        VGL_FRONTIER frontier(graph, SCATTER);
        graph_API.change_traversal_direction(SCATTER, widths, frontier);
        Timer tm;
        tm.start();
        const _T inf_val = std::numeric_limits<_T>::max() - MAX_WEIGHT;
        auto init_widths = [widths, source_vertex, inf_val] __VGL_COMPUTE_ARGS__ {
            widths[src_id] = (src_id == source_vertex) ? inf_val : (_T)0;
        };
        frontier.set_all_active();
        graph_API.compute(graph, frontier, init_widths);
        int *changes;
        MemoryAPI::allocate_array(&changes, 1);
        int iterations_count = 0;
        do {
            changes[0] = 0;
            iterations_count++;
            auto edge_op_push = [widths, edges_capacities, changes] __VGL_SCATTER_ARGS__ {
                const _T new_width = fminf(widths[src_id], edges_capacities[global_edge_pos]);      // vect_min(widths[src], edge_width)
                if (widths[dst_id] < new_width) {
                    atomicMax(reinterpret_cast<int *>(&widths[dst_id]), __float_as_int(new_width));
                    changes[0] = 1;
                }
            };
            graph_API.scatter(graph, frontier, edge_op_push);
        } while (changes[0]);
        MemoryAPI::free_array(changes);
        tm.end();
        performance_stats.print_algorithm_performance_stats("SSWP (Dijkstra, all-active, push, operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    static double hip_fused(VGL_Graph &graph, EdgesArray<float> &edges_capacities, VerticesArray<float> &widths, int source_vertex)
    {
        Timer tm;
        tm.start();
        vgl_hip_sssp_stats st;
        VGL_HIP_CALL(vgl_hip_sswp_run(VGL_RUNTIME::ctx(), graph.get_handle(), edges_capacities.get_ptr(), source_vertex, VGL_HIP_SSSP_ACTIVE_TILES,
                                      widths.get_ptr(), &st));
        tm.end();
        performance_stats.print_algorithm_performance_stats("SSWP (fused)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
#define SSWP WidestPaths
