// Speculative greedy colouring on the generic operator path.  Same primitives and the same order of calls per round as
// Coloring::vgl_coloring (algorithms/coloring/coloring.hpp:58-153) -- that is the point of this file: it shows that code written
// against the reference's operator API (sparse frontiers, a vertex post-op, a 64-bit VerticesArray, reduce<int>,
// generate_new_frontier in one loop) runs on GraphAbstractionsHIP.
//
// One round over the vertices that still need a colour (the frontier):
//   1. palette[v] = all 64 colours of the current window [base, base + 64)
//   2. scatter  : every edge v -> w strikes colour(w) out of palette[v] when it lies in the window   (atomic AND; the reference asks
//                 for enable_safe_stores on NEC for the same read-modify-write);  POST-op: colour(v) = lowest colour left, if any
//   3. scatter  : an edge whose endpoints ended up with equal colours marks its larger endpoint for another round
//   4. reduce   : did any vertex run out of colours in this window?  then the window moves up by 64
//   5. generate_new_frontier from the marks
// The outcome depends on the execution order (in the reference too); the app checks that the colouring is proper.
#pragma once

struct Coloring {
    typedef unsigned long long palette_t;
    static constexpr int WINDOW = 64;

    __host__ __device__ static inline int lowest_free(palette_t p)
    {
        int pos = -1;
        for (int b = 0; b < WINDOW && pos < 0; b++)
            if ((p >> b) & 1ULL) pos = b;
        return pos;
    }

    static double vgl_coloring(VGL_Graph &graph, VerticesArray<int> &colors, int *rounds_out = nullptr)
    {
        VerticesArray<palette_t> palette(graph);
        VerticesArray<int> again(graph);
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER todo(graph);
        api.change_traversal_direction(SCATTER, todo, colors, palette, again);
        Timer tm;
        tm.start();
        todo.set_all_active();
        api.compute(graph, todo, [colors] __VGL_COMPUTE_ARGS__ { colors[src_id] = 0; });
        int base = 0, rounds = 0;
        for (; todo.size() > 0; rounds++) {
            palette.set_all_constant(~(palette_t)0);
            again.set_all_constant(0);
            auto strike_out = [colors, palette, base] __VGL_SCATTER_ARGS__ {
                const int c = colors[dst_id] - base;
                if (src_id != dst_id && c >= 0 && c < WINDOW) atomicAnd(&palette[src_id], ~((palette_t)1 << c));
            };
            auto pick = [colors, palette, base] __VGL_ADVANCE_POSTPROCESS_ARGS__ {
                const int b = lowest_free(palette[src_id]);
                if (b >= 0) colors[src_id] = base + b;
            };
            api.enable_safe_stores();
            api.scatter(graph, todo, strike_out, EMPTY_VERTEX_OP, pick, strike_out, EMPTY_VERTEX_OP, pick);
            api.disable_safe_stores();
            auto find_conflicts = [colors, again] __VGL_SCATTER_ARGS__ {
                if (src_id != dst_id && colors[src_id] == colors[dst_id]) again[src_id < dst_id ? dst_id : src_id] = 1;
            };
            api.scatter(graph, todo, find_conflicts);
            const int exhausted = api.reduce<int>(graph, todo, [palette] __VGL_REDUCE_INT_ARGS__ { return palette[src_id] == 0 ? 1 : 0; }, REDUCE_SUM);
            if (exhausted > 0) base += WINDOW;
            api.generate_new_frontier(graph, todo, [again] __VGL_GNF_ARGS__ { return again[src_id] ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG; });
        }
        tm.end();
        if (rounds_out) *rounds_out = rounds;
        std::cout << "Iterations: " << rounds << std::endl;
        performance_stats.print_algorithm_performance_stats("Coloring (operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
