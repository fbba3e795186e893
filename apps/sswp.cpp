// sswp app: counterpart of apps/sswp/sswp.cpp:11-66 (random capacities, random non-isolated sources, -check against a
// sequential label-correcting run).
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/sswp/sswp.cpp:3)
#include "common.hpp"
#include "algorithms/sswp.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        VerticesArray<float> widths(graph, SCATTER);
        EdgesArray<float> capacities(graph);
        capacities.set_all_random(MAX_WEIGHT);
        double avg_perf = 0;
        for (int i = 0; i < parser.get_number_of_rounds(); i++) {
            const int source_vertex = graph.reorder(parser.source >= 0 ? checked_vertex(graph, parser.source, "source") : graph.select_random_nz_vertex(ORIGINAL, i), ORIGINAL, SCATTER);
            const double perf = parser.fused ? WidestPaths::hip_fused(graph, capacities, widths, source_vertex)
                                             : WidestPaths::vgl_dijkstra(graph, capacities, widths, source_vertex);
            avg_perf += perf / parser.get_number_of_rounds();
            if (parser.get_check_flag()) {
                HostCSR h(graph);
                verify_results(widths.to_host(), seq_widest_paths(h, capacities.outgoing_to_host(), source_vertex));
            }
        }
        widths.reorder(ORIGINAL);
        dump_array(parser.dump, widths.to_host());
        report_performance(avg_perf);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
