// HITS against the operator API: the call sequence of HITS::vgl_hits (algorithms/hits/hits.hpp:5-100) -- gather with a vertex
// pre-op that zeroes the accumulator and an edge-op that adds (VGL_SRC_ID_ADD = atomicAdd, as in the reference's GPU flavour,
// architecture_independent_api.h), reduce<double> of the squares, compute to normalise, then the same in the scatter direction.
// The f64 atomics make the last bits of this version order-dependent (like the reference's GPU variant); the fused path
// (vgl_hip_hits_run) keeps the sequential per-vertex order.
#pragma once

struct HITS {
    template <typename _T>
    static double vgl_hits(VGL_Graph &graph, VerticesArray<_T> &auth, VerticesArray<_T> &hub, int num_steps)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph);
        VGL_FRONTIER frontier(graph);
        graph_API.change_traversal_direction(GATHER, hub, auth, frontier);
        frontier.set_all_active();
        Timer tm;
        tm.start();
        auto init_op = [auth, hub] __VGL_COMPUTE_ARGS__ {
            auth[src_id] = 1;
            hub[src_id] = 1;
        };
        graph_API.compute(graph, frontier, init_op);
        for (int step = 0; step < num_steps; step++) {
            graph_API.change_traversal_direction(GATHER, hub, auth, frontier);
            auto update_auth_op_preprocess = [auth] __VGL_ADVANCE_PREPROCESS_ARGS__ { auth[src_id] = 0.0; };
            auto update_auth_op = [auth, hub] __VGL_ADVANCE_ARGS__ { VGL_SRC_ID_ADD(auth[src_id], hub[dst_id]); };
            graph_API.gather(graph, frontier, update_auth_op, update_auth_op_preprocess, EMPTY_VERTEX_OP,
                             update_auth_op, update_auth_op_preprocess, EMPTY_VERTEX_OP);
            auto reduce_auth_op = [auth] __VGL_REDUCE_DBL_ARGS__ { return auth[src_id] * auth[src_id]; };
            _T norm = sqrt(graph_API.reduce<_T>(graph, frontier, reduce_auth_op, REDUCE_SUM));
            auto normalize_auth_op = [auth, norm] __VGL_COMPUTE_ARGS__ { auth[src_id] /= norm; };
            graph_API.compute(graph, frontier, normalize_auth_op);

            graph_API.change_traversal_direction(SCATTER, hub, auth, frontier);
            auto update_hub_op_preprocess = [hub] __VGL_ADVANCE_PREPROCESS_ARGS__ { hub[src_id] = 0.0; };
            auto update_hub_op = [hub, auth] __VGL_ADVANCE_ARGS__ { VGL_SRC_ID_ADD(hub[src_id], auth[dst_id]); };
            graph_API.scatter(graph, frontier, update_hub_op, update_hub_op_preprocess, EMPTY_VERTEX_OP,
                              update_hub_op, update_hub_op_preprocess, EMPTY_VERTEX_OP);
            auto reduce_hub_op = [hub] __VGL_REDUCE_DBL_ARGS__ { return hub[src_id] * hub[src_id]; };
            norm = sqrt(graph_API.reduce<_T>(graph, frontier, reduce_hub_op, REDUCE_SUM));
            auto normalize_hub_op = [hub, norm] __VGL_COMPUTE_ARGS__ { hub[src_id] /= norm; };
            graph_API.compute(graph, frontier, normalize_hub_op);
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
