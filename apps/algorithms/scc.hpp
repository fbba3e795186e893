// Strongly connected components: the fused path (vgl_hip_scc_run) behind the interface of SCC::vgl_forward_backward
// (algorithms/scc/scc.hpp:268-300).  Labels are the smallest vertex id of each component -- a valid input for the reference's
// equal_components check, which only looks at the partition.
#pragma once

struct SCC {
    static double vgl_forward_backward(VGL_Graph &graph, VerticesArray<int> &components)
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
