// Strongly connected components.
//   vgl_forward_backward : against the operator API -- trim (vertices without a live in- or out-neighbour are components of their own, like the
//                          trim step of SCC::vgl_forward_backward, algorithms/scc/scc.hpp:62-124), then rounds of: forward propagation of the
//                          smallest id among the still-active vertices (scatter with atomicMin), backward reach from the vertices that kept
//                          their own id inside their colour class (the forward-backward step of scc.hpp:126-266 run for all classes at once).
//                          A vertex that keeps its own id is the smallest member of its component, so labels are smallest member ids.
//   hip_fused            : the library's fused path (vgl_hip_scc_run), same labels.
// Either output is a valid input of the reference's equal_components check, which only looks at the partition.
#pragma once

struct SCC {
    static double vgl_forward_backward(VGL_Graph &graph, VerticesArray<int> &components)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER live(graph);
        VerticesArray<int> active(graph, SCATTER), colours(graph, SCATTER), has_out(graph, SCATTER), has_in(graph, SCATTER), reached(graph, SCATTER);
        api.change_traversal_direction(SCATTER, components, live, active, colours, has_out, has_in, reached);
        Timer tm;
        tm.start();
        live.set_all_active();
        auto init = [components, active] __VGL_COMPUTE_ARGS__ { components[src_id] = -1; active[src_id] = 1; };
        api.compute(graph, live, init);
        vgl_device_words<1> flag;
        int *changed = flag.device();
        auto still_active = [active] __VGL_GNF_ARGS__ { return active[src_id] ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG; };
        int trim_rounds = 0, colour_rounds = 0, reach_rounds = 0, outer = 0;
        for (;;) {
            // ---- trim: repeat until every active vertex has an active in- and out-neighbour other than itself ----
            for (;;) {
                api.generate_new_frontier(graph, live, still_active);
                if (live.size() == 0) break;
                auto no_neighbours = [has_out, has_in] __VGL_COMPUTE_ARGS__ { has_out[src_id] = 0; has_in[src_id] = 0; };
                api.compute(graph, live, no_neighbours);
                auto live_edge = [active, has_out, has_in] __VGL_SCATTER_ARGS__ {
                    if (src_id != dst_id && active[dst_id]) { has_out[src_id] = 1; has_in[dst_id] = 1; }
                };
                api.scatter(graph, live, live_edge);
                flag.clear();
                auto trim = [components, active, has_out, has_in, changed] __VGL_COMPUTE_ARGS__ {
                    if (!has_out[src_id] || !has_in[src_id]) { components[src_id] = src_id; active[src_id] = 0; changed[0] = 1; }
                };
                api.compute(graph, live, trim);
                trim_rounds++;
                if (!flag.fetch(0)) break;
            }
            if (live.size() == 0) break;
            outer++;
            // ---- forward: every active vertex learns the smallest active id that reaches it ----
            auto own_colour = [colours, reached] __VGL_COMPUTE_ARGS__ { colours[src_id] = src_id; reached[src_id] = 0; };
            api.compute(graph, live, own_colour);
            for (;;) {
                flag.clear();
                auto spread = [active, colours, changed] __VGL_SCATTER_ARGS__ {
                    if (active[dst_id]) {
                        const int mine = colours[src_id];
                        if (mine < colours[dst_id]) { atomicMin(&colours[dst_id], mine); changed[0] = 1; }
                    }
                };
                api.scatter(graph, live, spread);
                colour_rounds++;
                if (!flag.fetch(0)) break;
            }
            // ---- backward: inside a colour class, whatever reaches the vertex that kept its own id is its component ----
            auto roots = [colours, reached] __VGL_COMPUTE_ARGS__ { reached[src_id] = colours[src_id] == src_id; };
            api.compute(graph, live, roots);
            for (;;) {
                flag.clear();
                auto back = [active, colours, reached, changed] __VGL_SCATTER_ARGS__ {      // edge src -> dst read from its tail: src joins when dst has
                    if (!reached[src_id] && active[dst_id] && reached[dst_id] && colours[dst_id] == colours[src_id]) { reached[src_id] = 1; changed[0] = 1; }
                };
                api.scatter(graph, live, back);
                reach_rounds++;
                if (!flag.fetch(0)) break;
            }
            auto settle = [components, active, colours, reached] __VGL_COMPUTE_ARGS__ {
                if (reached[src_id]) { components[src_id] = colours[src_id]; active[src_id] = 0; }
            };
            api.compute(graph, live, settle);
        }
        tm.end();
        std::cout << "trim rounds: " << trim_rounds << ", forward-backward steps: " << outer << ", colour rounds: " << colour_rounds << ", reach rounds: "
                  << reach_rounds << std::endl;
        performance_stats.print_algorithm_performance_stats("SCC (trim + colour forward-backward, operator API)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }

    static double hip_fused(VGL_Graph &graph, VerticesArray<int> &components)
    {
        Timer tm;
        tm.start();
        vgl_hip_scc_stats st;
        VGL_HIP_CALL(vgl_hip_scc_run(VGL_RUNTIME::ctx(), graph.get_handle(), components.get_ptr(), &st));
        tm.end();
        std::cout << "trim rounds: " << st.trim_rounds << ", forward-backward steps: " << st.forward_backward_steps << ", colour rounds: "
                  << st.colour_rounds << " (" << st.edge_passes << " edge passes)" << std::endl;
        performance_stats.print_algorithm_performance_stats("SCC (trim + forward-backward + colouring, fused)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
