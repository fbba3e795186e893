// Shiloach-Vishkin against the operator API (call sequence of CC::vgl_shiloach_vishkin, algorithms/cc/shiloach_vishkin.hpp:7-88,
// flags in host-visible memory like gpu_shiloach_vishkin.hpp:30-68).
#pragma once

struct ConnectedComponents {
    // declared = true: the hook is handed over as a DECLARED operator (VGL_MIN_LABEL_OVER_EDGES, an extension of the API: the backend runs it
    // as its blocked pass) instead of the lambda with atomicMin; the pointer jumps stay lambdas.  Same labels.
    template <typename _T>
    static double vgl_shiloach_vishkin(VGL_Graph &graph, VerticesArray<_T> &components, bool declared = false)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER all(graph);
        api.change_traversal_direction(SCATTER, components, all);
        if (declared)                    // the layout behind the declared hook: once per graph, outside the run (vgl_hip_cc_prepare)
            std::cout << "CC declared-hook layout (blocked adjacency, once per graph): " << 1000.0 * api.prepare(graph, VGL_MIN_LABEL_OVER_EDGES(components)) << " ms" << std::endl;
        Timer tm;
        tm.start();
        all.set_all_active();
        auto own_id = [components] __VGL_COMPUTE_ARGS__ { components[src_id] = src_id; };
        api.compute(graph, all, own_id);
        // the two loop flags live in device memory (vgl_device_words): the operators' stores stay on the card and no primitive has to end
        // with a synchronisation; the host fetches a flag where the reference reads its managed word
        vgl_device_words<2> flags;
        int *hooked = flags.device(), *jumped = flags.device() + 1;
        do {
            flags.clear();
            auto hook = [components, hooked] __VGL_SCATTER_ARGS__ {
                const int label = components[src_id];
                if (label < components[dst_id]) { atomicMin(&components[dst_id], label); hooked[0] = 1; }
            };
            int any_hook;
            if (declared) any_hook = api.scatter(graph, all, VGL_MIN_LABEL_OVER_EDGES(components)) ? 1 : 0;
            else { api.scatter(graph, all, hook); any_hook = flags.fetch(0); }
            int any_jump;
            do {
                flags.clear();
                auto shortcut = [components, jumped] __VGL_COMPUTE_ARGS__ {
                    const int label = components[src_id];
                    const int grand = components[label];
                    if (label != grand) { components[src_id] = grand; jumped[0] = 1; }
                };
                api.compute(graph, all, shortcut);
                any_jump = flags.fetch(1);
            } while (any_jump);
            if (!any_hook) break;
        } while (true);
        tm.end();
        performance_stats.print_algorithm_performance_stats(declared ? "CC (Shiloach-Vishkin, operator API, declared hook)" : "CC (Shiloach-Vishkin, operator API)", tm.get_time(), graph.get_edges_count());
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
