// bfs app: counterpart of apps/bfs/bfs.cpp:15-62 (rounds with random non-isolated sources, optional -check).
#define INT_ELEMENTS_PER_EDGE 4.0      // VGL byte accounting of this app (apps/bfs/bfs.cpp:3)
#include "common.hpp"
#include "algorithms/bfs.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        if (parser.blocked) {
            if (!parser.fused) throw "-blocked goes with -fused (the blocked levels are part of vgl_hip_bfs_run)";
            VGL_HIP_CALL(vgl_hip_bfs_prepare_blocked(VGL_RUNTIME::ctx(), graph.get_handle()));     // graph preparation, like the import: outside the timers
        }
        VerticesArray<int> levels(graph, SCATTER);
        double avg_perf = 0;
        for (int i = 0; i < parser.get_number_of_rounds(); i++) {
            // -source and select_random_nz_vertex speak ORIGINAL ids; algorithms and checkers work in the stored numbering
            const int source_vertex = graph.reorder(parser.source >= 0 ? checked_vertex(graph, parser.source, "source") : graph.select_random_nz_vertex(ORIGINAL, i), ORIGINAL, SCATTER);
            const double perf = parser.fused ? BFS::hip_fused(graph, levels, source_vertex, parser.direction_optimising)
                                             : BFS::vgl_top_down(graph, levels, source_vertex);
            avg_perf += perf / parser.get_number_of_rounds();
            if (parser.get_check_flag()) {
                HostCSR h(graph);
                verify_results(levels.to_host(), seq_bfs(h, source_vertex));
            }
        }
        levels.reorder(ORIGINAL);
        dump_array(parser.dump, levels.to_host());
        report_performance(avg_perf);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
