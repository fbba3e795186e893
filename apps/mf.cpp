// mf app: counterpart of apps/mf/mf.cpp:11-70 (rounds with a source / sink pair each; -source / -sink fix them; -undirected generates the
// symmetric input on which the result is the maximum flow; -check compares with a sequential run of the same rules on the host).
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/mf/mf.cpp:3)
#include "common.hpp"
#include "algorithms/mf.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, parser.undirected ? UNDIRECTED_GRAPH : DIRECTED_GRAPH);
        EdgesArray<int> flows(graph);
        double avg_perf = 0;
        std::vector<int> results;
        for (int i = 0; i < parser.get_number_of_rounds(); i++) {
            flows.set_all_constant(MAX_WEIGHT);                  // clear flows before each round (mf.cpp:36)
            int max_flow_val = 0;
            const int source_original = parser.source >= 0 ? checked_vertex(graph, parser.source, "source") : graph.select_random_nz_vertex(ORIGINAL, 2 * i);
            const int sink_original = parser.sink >= 0 ? checked_vertex(graph, parser.sink, "sink") : graph.select_random_nz_vertex(ORIGINAL, 2 * i + 1);
            const int source = graph.reorder(source_original, ORIGINAL, SCATTER), sink = graph.reorder(sink_original, ORIGINAL, SCATTER);
            avg_perf += MF::vgl_ford_fulkerson(graph, flows, source, sink, max_flow_val) / parser.get_number_of_rounds();
            std::cout << "Result: " << max_flow_val << " (source " << source_original << ", sink " << sink_original << ")" << std::endl;
            results.push_back(source_original); results.push_back(sink_original); results.push_back(max_flow_val);
            if (parser.get_check_flag()) {
                HostCSR h(graph);
                const int check_flow = seq_ford_fulkerson(h, source, sink, MAX_WEIGHT);
                std::cout << max_flow_val << " vs " << check_flow << std::endl;
                std::cout << "error count: " << (max_flow_val == check_flow ? 0 : graph.get_vertices_count()) << std::endl;
            }
        }
        dump_array(parser.dump, results);
        report_performance(avg_perf);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
