// PageRank against the operator API (call sequence of PR::vgl_page_rank, algorithms/pr/pr.hpp:7-149, in the shape of its
// GPU variant gpu_pr.hpp:100-175: gather-free setup over the outgoing graph, per iteration compute / reduce / scatter with
// atomic accumulation into rank[src] / compute).
#pragma once

struct PageRank {
    template <typename _T>
    // deterministic: 0 = float atomics (gpu_pr.hpp's shape), 1 = sequential rows (one lane per vertex walks its edges between pre and post),
    // 2 = pull inside compute(): the lane of a vertex sums its neighbours' contributions in a register (adjacency through a vgl_csr_view
    // captured by value, as the reference's random_walk.hpp reads the graph inside compute lambdas).  1 and 2 are the multicore recipe
    // (pr.hpp:105-124): products and sums in f32 in adjacency order -- seq_page_rank's chain, independent of the schedule.
    // 3 = the pull as a DECLARED operator (VGL_SUM_OVER_EDGES, an API extension): the class runs the library's blocked pass -- exact sums, the same
    // bits for any schedule, <= 1e-6 of the chain while rows are short (uniform graphs).
    static double vgl_page_rank(VGL_Graph &graph, VerticesArray<_T> &page_ranks, _T, int max_iterations, int deterministic = 0)
    {
        const int vertices_count = graph.get_vertices_count();
        VGL_GRAPH_ABSTRACTIONS graph_API(graph);
        VGL_FRONTIER frontier(graph);
        VerticesArray<int> incoming_degrees_without_loops(graph, SCATTER);
        VerticesArray<_T> reversed_degrees(graph, SCATTER);
        VerticesArray<_T> old_page_ranks(graph, SCATTER);
        VerticesArray<_T> contributions(graph, SCATTER);        // old rank * reversed degree, once per vertex instead of once per edge (the same f32 product)
        graph_API.change_traversal_direction(SCATTER, frontier, incoming_degrees_without_loops, reversed_degrees, old_page_ranks, contributions, page_ranks);
        frontier.set_all_active();
        const _T d = 0.85;
        const _T k = (1.0 - d) / ((_T)vertices_count);
        auto init_data = [page_ranks, incoming_degrees_without_loops, vertices_count] __VGL_COMPUTE_ARGS__ {
            page_ranks[src_id] = 1.0 / vertices_count;
            incoming_degrees_without_loops[src_id] = 0;
        };
        graph_API.compute(graph, frontier, init_data);
        auto count_incoming = [incoming_degrees_without_loops] __VGL_SCATTER_ARGS__ {
            if (src_id != dst_id) atomicAdd(&incoming_degrees_without_loops[dst_id], 1);
        };
        graph_API.scatter(graph, frontier, count_incoming);
        if (vgl_library_data.get_mpi_proc_num() > 1) {      // every rank counted the edges of its vertex range (the exchange at pr.hpp:58)
            auto sum_op = [] __device__ (int a, int b) -> int { return a + b; };
            graph_API.exchange_vertices_array(EXCHANGE_ALL, graph, incoming_degrees_without_loops, sum_op);
        }
        auto calculate_reversed_degrees = [reversed_degrees, incoming_degrees_without_loops] __VGL_COMPUTE_ARGS__ {
            const int dg = incoming_degrees_without_loops[src_id];
            reversed_degrees[src_id] = (dg == 0) ? (_T)0 : (_T)(1.0 / dg);
        };
        graph_API.compute(graph, frontier, calculate_reversed_degrees);
        if (deterministic == 3)          // the layout behind the declared sum: once per graph, outside the run like the library's own plan (vgl_hip_pr_prepare)
            std::cout << "PR declared-sum layout (blocked adjacency, once per graph): " << 1000.0 * graph_API.prepare(graph, VGL_SUM_OVER_EDGES(page_ranks, contributions)) << " ms" << std::endl;
        Timer tm;
        tm.start();
        for (int it = 0; it < max_iterations; it++) {
            auto save_old_ranks = [old_page_ranks, page_ranks, contributions, reversed_degrees] __VGL_COMPUTE_ARGS__ {
                old_page_ranks[src_id] = page_ranks[src_id];
                contributions[src_id] = page_ranks[src_id] * reversed_degrees[src_id];
                page_ranks[src_id] = 0;
            };
            graph_API.compute(graph, frontier, save_old_ranks);
            auto reduce_dangling_input = [incoming_degrees_without_loops, old_page_ranks, vertices_count] __VGL_REDUCE_FLT_ARGS__ {
                return incoming_degrees_without_loops[src_id] == 0 ? old_page_ranks[src_id] / vertices_count : 0.0f;
            };
            const _T dangling_input = (_T)graph_API.template reduce<double>(graph, frontier, reduce_dangling_input, REDUCE_SUM);
            auto vertex_postprocess_op = [page_ranks, k, d, dangling_input] __VGL_ADVANCE_POSTPROCESS_ARGS__ {
                page_ranks[src_id] = k + d * (page_ranks[src_id] + dangling_input);
            };
            if (deterministic == 3) {
                graph_API.scatter(graph, frontier, VGL_SUM_OVER_EDGES(page_ranks, contributions), EMPTY_VERTEX_OP, vertex_postprocess_op);
            } else if (deterministic == 2) {
                const vgl_csr_view out = graph.get_direction_view(SCATTER);
                auto pull = [page_ranks, contributions, out, k, d, dangling_input] __VGL_COMPUTE_ARGS__ {
                    _T sum = 0;
                    for (long long e = out.rowptr[src_id]; e < out.rowptr[src_id + 1]; e++) {
                        const int dst_id = out.adj[e];
                        if (src_id != dst_id) sum = sum + contributions[dst_id];
                    }
                    page_ranks[src_id] = k + d * (sum + dangling_input);
                };
                graph_API.compute(graph, frontier, pull);
            } else if (deterministic == 1) {
                auto edge_op_seq = [page_ranks, contributions] __VGL_SCATTER_ARGS__ {
                    if (src_id != dst_id) page_ranks[src_id] = page_ranks[src_id] + contributions[dst_id];
                };
                graph_API.enable_sequential_rows();
                graph_API.scatter(graph, frontier, edge_op_seq, EMPTY_VERTEX_OP, vertex_postprocess_op, edge_op_seq, EMPTY_VERTEX_OP, vertex_postprocess_op);
                graph_API.disable_sequential_rows();
            } else {
                auto edge_op = [page_ranks, contributions] __VGL_SCATTER_ARGS__ {
                    if (src_id != dst_id) VGL_SRC_ID_ADD(page_ranks[src_id], contributions[dst_id]);
                };
                graph_API.scatter(graph, frontier, edge_op, EMPTY_VERTEX_OP, vertex_postprocess_op, edge_op, EMPTY_VERTEX_OP, vertex_postprocess_op);
            }
            graph_API.exchange_vertices_array(EXCHANGE_PRIVATE_DATA, graph, page_ranks);       // pr.hpp:127: the ranks of every owner's vertex range
        }
        tm.end();
        auto reduce_ranks_sum = [page_ranks] __VGL_REDUCE_FLT_ARGS__ { return page_ranks[src_id]; };
        std::cout << "ranks sum: " << graph_API.template reduce<double>(graph, frontier, reduce_ranks_sum, REDUCE_SUM) << std::endl;
        performance_stats.print_algorithm_performance_stats("PR (operator API)", tm.get_time(), graph.get_edges_count());
        return max_iterations * performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    static double hip_fused(VGL_Graph &graph, VerticesArray<float> &page_ranks, int max_iterations)
    {
        Timer tm;
        tm.start();
        vgl_hip_pr_stats st;
        VGL_HIP_CALL(vgl_hip_pr_run(VGL_RUNTIME::ctx(), graph.get_handle(), nullptr, max_iterations, page_ranks.get_ptr(), &st));
        tm.end();
        std::cout << "ranks sum: " << st.ranks_sum << std::endl;
        performance_stats.print_algorithm_performance_stats("PR (fused)", tm.get_time(), graph.get_edges_count());
        return max_iterations * performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
#define PR PageRank
