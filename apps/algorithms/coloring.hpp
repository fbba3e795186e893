// Greedy speculative graph colouring against the operator API: the call sequence of Coloring::vgl_coloring
// (algorithms/coloring/coloring.hpp:58-153).  Every round: the vertices still to be coloured mark the colours of their neighbours
// in a 64-colour window as forbidden (scatter with a vertex POST-op that picks the smallest free colour), conflicts between
// neighbours are detected (second scatter) and the larger endpoint of every conflict goes into the next frontier
// (generate_new_frontier); reduce<int> decides whether the colour window moves on.  The result depends on the execution order (in the
// reference as well): the app verifies that it is a PROPER colouring, there is no single expected output.
// The forbidden-colour update uses an atomic AND (the reference needs enable_safe_stores on NEC for the same read-modify-write).
#pragma once

struct Coloring {
    __host__ __device__ static inline int smallest_bit_pos(unsigned long long x)
    {
        if (x == 0) return -1;
        int pos = 0;
        while (!(x & 1ULL)) { x >>= 1; pos++; }
        return pos;
    }

    static double vgl_coloring(VGL_Graph &graph, VerticesArray<int> &colors, int *rounds_out = nullptr)
    {
        VerticesArray<unsigned long long> available_colors(graph);
        VerticesArray<int> need_recolor(graph);
        VGL_GRAPH_ABSTRACTIONS graph_API(graph);
        VGL_FRONTIER frontier(graph);
        graph_API.change_traversal_direction(SCATTER, frontier, colors, available_colors, need_recolor);
        Timer tm;
        tm.start();
        frontier.set_all_active();
        auto init_op = [colors, available_colors] __VGL_COMPUTE_ARGS__ {
            colors[src_id] = 0;
            available_colors[src_id] = ~0ULL;
        };
        graph_API.compute(graph, frontier, init_op);
        int start_range = 0, end_range = 64, iterations = 0;
        while (frontier.size() > 0) {
            available_colors.set_all_constant(~0ULL);
            auto mark_forbidden_op = [colors, available_colors, start_range, end_range] __VGL_SCATTER_ARGS__ {
                const int dst_color = colors[dst_id];
                if (dst_color >= start_range && dst_color < end_range && src_id != dst_id)
                    atomicAnd(&available_colors[src_id], ~(1ULL << (dst_color - start_range)));
            };
            auto vertex_postprocess_op = [colors, available_colors, start_range] __VGL_ADVANCE_POSTPROCESS_ARGS__ {
                const int bit_pos = smallest_bit_pos(available_colors[src_id]);
                if (bit_pos >= 0) colors[src_id] = bit_pos + start_range;
            };
            graph_API.enable_safe_stores();
            graph_API.scatter(graph, frontier, mark_forbidden_op, EMPTY_VERTEX_OP, vertex_postprocess_op, mark_forbidden_op, EMPTY_VERTEX_OP,
                              vertex_postprocess_op);
            graph_API.disable_safe_stores();
            need_recolor.set_all_constant(0);
            auto create_reordering_op = [colors, need_recolor] __VGL_SCATTER_ARGS__ {
                if (colors[dst_id] == colors[src_id] && src_id != dst_id) need_recolor[src_id > dst_id ? src_id : dst_id] = 1;
            };
            graph_API.scatter(graph, frontier, create_reordering_op);
            auto offset_change_required_op = [available_colors] __VGL_REDUCE_INT_ARGS__ { return available_colors[src_id] == 0 ? 1 : 0; };
            const int full_vertices = graph_API.reduce<int>(graph, frontier, offset_change_required_op, REDUCE_SUM);
            if (full_vertices > 0) { start_range += 64; end_range += 64; }
            auto need_recolor_op = [need_recolor] __VGL_GNF_ARGS__ { return need_recolor[src_id] == 1 ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG; };
            graph_API.generate_new_frontier(graph, frontier, need_recolor_op);
            iterations++;
        }
        tm.end();
        if (rounds_out) *rounds_out = iterations;
        std::cout << "Iterations: " << iterations << std::endl;
        performance_stats.print_algorithm_performance_stats("Coloring (operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
