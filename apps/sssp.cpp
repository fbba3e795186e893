// sssp app: counterpart of apps/sssp/sssp.cpp:16-80.
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/sssp/sssp.cpp:3)
#include "common.hpp"
#include "algorithms/sssp.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        VerticesArray<float> distances(graph, SCATTER);
        EdgesArray<float> weights(graph);
        weights.set_all_random(MAX_WEIGHT);
        double avg_perf = 0;
        for (int i = 0; i < parser.get_number_of_rounds(); i++) {
            const int source_vertex = graph.reorder(parser.source >= 0 ? checked_vertex(graph, parser.source, "source") : graph.select_random_nz_vertex(ORIGINAL, i), ORIGINAL, SCATTER);
            // SSSP::vgl_dijkstra's dispatch (shortest_paths.hpp:296-318): -push / -pull select the traversal; -fused takes the library's
            // schedules (delta-stepping; with -pull the blocked pull steps, with -do push <-> pull switching)
            const bool pull = parser.traversal == Parser::PULL_TRAVERSAL;
            const double perf = parser.fused ? ((pull || parser.direction_optimising) ? ShortestPaths::hip_fused_pull(graph, weights, distances, source_vertex, parser.direction_optimising)
                                                                                      : ShortestPaths::hip_fused(graph, weights, distances, source_vertex))
                                             : pull ? ShortestPaths::vgl_dijkstra_all_active_pull(graph, weights, distances, source_vertex)
                                                    : ShortestPaths::vgl_dijkstra_all_active_push(graph, weights, distances, source_vertex, parser.declared);
            avg_perf += perf / parser.get_number_of_rounds();
            if (parser.get_check_flag()) {
                HostCSR h(graph);
                verify_results(distances.to_host(), seq_dijkstra(h, weights.outgoing_to_host(), source_vertex));
            }
        }
        distances.reorder(ORIGINAL);
        dump_array(parser.dump, distances.to_host());
        report_performance(avg_perf);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
