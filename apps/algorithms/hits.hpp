// HITS on the operator API.  The primitives and their order are those of HITS::vgl_hits (algorithms/hits/hits.hpp:5-100): per
// step, in the GATHER direction authorities are rebuilt from the hubs of the in-neighbours (vertex pre-op zeroes, edge-op adds with
// VGL_SRC_ID_ADD = atomicAdd as in the reference's GPU flavour), reduce<double> of the squares + compute normalise; then the same in
// the SCATTER direction for the hubs.  The two halves differ only in the direction and in which array is read / written, so they
// share one helper here.  The f64 atomics make the last bits of this version order-dependent (like the reference's GPU variant);
// the fused path (vgl_hip_hits_run) keeps the sequential per-vertex order.
#pragma once

struct HITS {
    // target[v] = sum over v's neighbours (in the current traversal direction) of source[neighbour], then target /= ||target||_2
    template <typename _T, typename Advance>
    static void half_step(VGL_GRAPH_ABSTRACTIONS &api, VGL_Graph &graph, VGL_FRONTIER &all, VerticesArray<_T> &target, VerticesArray<_T> &source,
                          Advance &&advance)
    {
        auto zero = [target] __VGL_ADVANCE_PREPROCESS_ARGS__ { target[src_id] = 0.0; };
        auto accumulate = [target, source] __VGL_ADVANCE_ARGS__ { VGL_SRC_ID_ADD(target[src_id], source[dst_id]); };
        advance(accumulate, zero);
        const _T norm = sqrt(api.template reduce<_T>(graph, all, [target] __VGL_REDUCE_DBL_ARGS__ { return target[src_id] * target[src_id]; }, REDUCE_SUM));
        api.compute(graph, all, [target, norm] __VGL_COMPUTE_ARGS__ { target[src_id] /= norm; });
    }

    template <typename _T>
    static double vgl_hits(VGL_Graph &graph, VerticesArray<_T> &auth, VerticesArray<_T> &hub, int num_steps)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER all(graph);
        api.change_traversal_direction(GATHER, hub, auth, all);
        all.set_all_active();
        Timer tm;
        tm.start();
        api.compute(graph, all, [auth, hub] __VGL_COMPUTE_ARGS__ { auth[src_id] = 1; hub[src_id] = 1; });
        for (int step = 0; step < num_steps; step++) {
            api.change_traversal_direction(GATHER, hub, auth, all);
            half_step(api, graph, all, auth, hub, [&](auto &edge_op, auto &pre_op) {
                api.gather(graph, all, edge_op, pre_op, EMPTY_VERTEX_OP, edge_op, pre_op, EMPTY_VERTEX_OP);
            });
            api.change_traversal_direction(SCATTER, hub, auth, all);
            half_step(api, graph, all, hub, auth, [&](auto &edge_op, auto &pre_op) {
                api.scatter(graph, all, edge_op, pre_op, EMPTY_VERTEX_OP, edge_op, pre_op, EMPTY_VERTEX_OP);
            });
        }
        tm.end();
        performance_stats.print_algorithm_performance_stats("VGL HITS (operator API)", tm.get_time(), graph.get_edges_count());
        return num_steps * performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    static double hip_fused(VGL_Graph &graph, VerticesArray<double> &auth, VerticesArray<double> &hub, int num_steps)
    {
        Timer tm;
        tm.start();
        VGL_HIP_CALL(vgl_hip_hits_run(VGL_RUNTIME::ctx(), graph.get_handle(), num_steps, auth.get_ptr(), hub.get_ptr()));
        tm.end();
        performance_stats.print_algorithm_performance_stats("HITS (fused)", tm.get_time(), graph.get_edges_count());
        return num_steps * performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
