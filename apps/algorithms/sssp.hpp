// All-active push Bellman-Ford against the operator API (call sequence of SSSP::vgl_dijkstra_all_active_push,
// algorithms/sssp/shortest_paths.hpp:85-163; the `changes` word lives in host-visible memory like the reference's GPU
// variant, gpu_shortest_paths.hpp:92-113).  The relaxation uses an integer atomic-min on the f32 bits so that no update
// is lost; the fixed point is the same bit pattern either way.
#pragma once

struct ShortestPaths {
    template <typename _T>
    // declared = true: the relax is handed over as a DECLARED operator (VGL_RELAX_OVER_EDGES, an extension of the API: the class runs it as a
    // blocked pass over a layout it builds once per weight array) instead of the lambda with atomicMin.  Same distances, bit for bit.
    static double vgl_dijkstra_all_active_push(VGL_Graph &graph, EdgesArray<_T> &weights, VerticesArray<_T> &distances, int source_vertex, bool declared = false)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph, SCATTER);
        VGL_FRONTIER frontier(graph, SCATTER);
        graph_API.change_traversal_direction(SCATTER, distances, frontier);
        if (declared && vgl_library_data.get_mpi_proc_num() == 1) {      // the layout behind the declared relax: once per weights, outside the run like the fused path's plan
            Timer tp;
            tp.start();
            graph.get_relax_plan(weights.get_ptr(), weights.version());
            tp.end();
            tp.print_time_stats("SSSP declared-relax layout (blocked adjacency + weights, once per weights)");
        }
        Timer tm;
        tm.start();
        const _T inf_val = std::numeric_limits<_T>::max() - MAX_WEIGHT;
        auto init_distances = [distances, source_vertex, inf_val] __VGL_COMPUTE_ARGS__ {
            distances[src_id] = (src_id == source_vertex) ? (_T)0 : inf_val;
        };
        frontier.set_all_active();
        graph_API.compute(graph, frontier, init_distances);
        vgl_device_words<1> changes_word;                     // the reference's `changes[0]`, kept on the card (vgl_device_words)
        int *changes = changes_word.device();
        int any_change = 0;
        int iterations_count = 0;
        // several ranks (the reference's __USE_MPI__ flavour, shortest_paths.hpp:112-154): every rank relaxes the edges of its vertex range,
        // the copies of the distances are merged through exchange_vertices_array(EXCHANGE_RECENTLY_CHANGED, ..., min_op) and the loop
        // condition is a reduce over the merged (replicated) arrays, so that every rank takes the same decision
        const bool several_ranks = vgl_library_data.get_mpi_proc_num() > 1;
        VerticesArray<_T> prev_distances(graph, SCATTER);
        do {
            changes_word.clear();
            iterations_count++;
            if (several_ranks) {
                auto save_old_distances = [prev_distances, distances] __VGL_COMPUTE_ARGS__ { prev_distances[src_id] = distances[src_id]; };
                graph_API.compute(graph, frontier, save_old_distances);
            }
            auto edge_op_push = [distances, weights, changes, inf_val] __VGL_SCATTER_ARGS__ {
                const _T src_weight = distances[src_id];
                if (src_weight < inf_val) {
                    const _T candidate = __fadd_rn(src_weight, weights[global_edge_pos]);
                    if (distances[dst_id] > candidate) {
                        atomicMin(reinterpret_cast<int *>(&distances[dst_id]), __float_as_int(candidate));
                        changes[0] = 1;
                    }
                }
            };
            bool declared_change = false;
            if (declared && !several_ranks) declared_change = graph_API.scatter(graph, frontier, VGL_RELAX_OVER_EDGES(distances, weights));
            else graph_API.scatter(graph, frontier, edge_op_push);
            if (several_ranks) {
                auto min_op = [] __device__ (_T a, _T b) -> _T { return a < b ? a : b; };
                graph_API.exchange_vertices_array(EXCHANGE_RECENTLY_CHANGED, graph, distances, prev_distances, min_op);      // shortest_paths.hpp:136-141
                auto reduce_changes = [prev_distances, distances] __VGL_REDUCE_INT_ARGS__ { return prev_distances[src_id] != distances[src_id] ? 1 : 0; };
                any_change = graph_API.template reduce<int>(graph, frontier, reduce_changes, REDUCE_SUM) > 0;                 // shortest_paths.hpp:143-152
            } else any_change = (declared ? (declared_change ? 1 : 0) : changes_word.fetch(0));
        } while (any_change);
        tm.end();
        performance_stats.print_algorithm_performance_stats(declared ? "SSSP (Bellman-Ford, all-active, operator API, declared relax)" : "SSSP (Bellman-Ford, all-active, push, operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    // All-active pull (call sequence of SSSP::vgl_dijkstra_all_active_pull, shortest_paths.hpp:169-292): every vertex takes the minimum
    // over its INCOMING edges; the weights are read from the incoming half of the EdgesArray through global_edge_pos.  The reference
    // keeps a per-thread register array between pre / edge / post operators; a device lambda folds into the vertex's own slot with an
    // integer atomic-min instead (only lanes of the same row meet there), which needs neither pre nor post operator.
    template <typename _T>
    static double vgl_dijkstra_all_active_pull(VGL_Graph &graph, EdgesArray<_T> &weights, VerticesArray<_T> &distances, int source_vertex)
    {
        VGL_GRAPH_ABSTRACTIONS graph_API(graph, GATHER);
        VGL_FRONTIER frontier(graph, GATHER);
        graph_API.change_traversal_direction(GATHER, distances, frontier);
        Timer tm;
        tm.start();
        const _T inf_val = std::numeric_limits<_T>::max() - MAX_WEIGHT;
        auto init_distances = [distances, source_vertex, inf_val] __VGL_COMPUTE_ARGS__ {
            distances[src_id] = (src_id == source_vertex) ? (_T)0 : inf_val;
        };
        frontier.set_all_active();
        graph_API.compute(graph, frontier, init_distances);
        int *changes;
        MemoryAPI::allocate_array(&changes, 1);
        do {
            changes[0] = 0;
            auto edge_op_pull = [distances, weights, changes, inf_val] __VGL_GATHER_ARGS__ {
                const _T dst_weight = distances[dst_id];
                if (dst_weight < inf_val) {
                    const _T candidate = __fadd_rn(dst_weight, weights[global_edge_pos]);
                    if (distances[src_id] > candidate) {
                        atomicMin(reinterpret_cast<int *>(&distances[src_id]), __float_as_int(candidate));
                        changes[0] = 1;
                    }
                }
            };
            graph_API.gather(graph, frontier, edge_op_pull);
        } while (changes[0]);
        MemoryAPI::free_array(changes);
        tm.end();
        graph_API.change_traversal_direction(SCATTER, distances, frontier);
        performance_stats.print_algorithm_performance_stats("SSSP (Bellman-Ford, all-active, pull, operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    // fused pull / direction-optimising schedule of libvgl_hip.so (blocked pull steps, vgl_hip_sssp_run_pull): the blocked copy of
    // (adjacency, weights) is built once per EdgesArray, its build time is printed, not charged to the traversals
    static double hip_fused_pull(VGL_Graph &graph, EdgesArray<float> &weights, VerticesArray<float> &distances, int source_vertex, bool direction_optimising)
    {
        // the plan holds a reordered copy of the weights: rebuilt when the graph handle, the array or its contents (EdgesArray::version,
        // bumped by set_all_random / set_all_constant) changed
        static vgl_hip_sssp_pull_plan *plan = nullptr;
        static const void *plan_graph = nullptr, *plan_weights = nullptr;
        static unsigned long long plan_version = 0;
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        if (!plan || plan_graph != (const void *)graph.get_handle() || plan_weights != (const void *)weights.get_ptr() || plan_version != weights.version()) {
            if (plan) vgl_hip_sssp_pull_plan_destroy(c, plan);
            plan = nullptr;
            Timer tp;
            tp.start();
            VGL_HIP_CALL(vgl_hip_sssp_pull_plan_create(c, graph.get_handle(), weights.get_ptr(), &plan));
            tp.end();
            tp.print_time_stats("SSSP pull plan (blocked adjacency + weights, once per weights)");
            plan_graph = graph.get_handle(); plan_weights = weights.get_ptr(); plan_version = weights.version();
        }
        Timer tm;
        tm.start();
        vgl_hip_sssp_stats st;
        VGL_HIP_CALL(vgl_hip_sssp_run_pull(c, graph.get_handle(), weights.get_ptr(), plan, source_vertex,
                                           direction_optimising ? VGL_HIP_SSSP_DIRECTION_OPT : VGL_HIP_SSSP_PULL, distances.get_ptr(), &st));
        tm.end();
        performance_stats.print_algorithm_performance_stats(direction_optimising ? "SSSP (fused, push <-> pull)" : "SSSP (fused, pull)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    // fused fast path of libvgl_hip.so: delta-stepping over a light / heavy split of the adjacency (same distances bit for bit).  The
    // split ("plan") depends on the weights only, so it is built once per EdgesArray and reused by every source; its build time is
    // printed, not charged to the traversals.
    static double hip_fused(VGL_Graph &graph, EdgesArray<float> &weights, VerticesArray<float> &distances, int source_vertex, float delta = 10.0f)
    {
        static vgl_hip_sssp_plan *plan = nullptr;
        static const void *plan_graph = nullptr, *plan_weights = nullptr;
        static float plan_delta = 0.0f;
        static unsigned long long plan_version = 0;
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        if (!plan || plan_graph != (const void *)graph.get_handle() || plan_weights != (const void *)weights.get_ptr() || plan_delta != delta ||
            plan_version != weights.version()) {
            if (plan) vgl_hip_sssp_plan_destroy(c, plan);
            plan = nullptr;
            Timer tp;
            tp.start();
            VGL_HIP_CALL(vgl_hip_sssp_plan_create(c, graph.get_handle(), weights.get_ptr(), delta, &plan));
            tp.end();
            tp.print_time_stats("SSSP plan (light / heavy split, once per weights)");
            plan_graph = graph.get_handle(); plan_weights = weights.get_ptr(); plan_delta = delta; plan_version = weights.version();
        }
        Timer tm;
        tm.start();
        vgl_hip_sssp_stats st;
        VGL_HIP_CALL(vgl_hip_sssp_run_plan(c, graph.get_handle(), plan, source_vertex, distances.get_ptr(), &st));
        tm.end();
        performance_stats.print_algorithm_performance_stats("SSSP (fused, delta-stepping)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
#define SSSP ShortestPaths
