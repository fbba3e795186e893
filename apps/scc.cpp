// scc app: counterpart of apps/scc/scc.cpp:11-52 (directed input; -check against a sequential Tarjan, partitions compared).
#define INT_ELEMENTS_PER_EDGE 4.0      // VGL byte accounting of this app (apps/scc/scc.cpp:3)
#include "common.hpp"
#include "algorithms/scc.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, DIRECTED_GRAPH);
        VerticesArray<int> components(graph, SCATTER);
        auto run = [&]() { return parser.fused ? SCC::hip_fused(graph, components) : SCC::vgl_forward_backward(graph, components); };
        run();                                                   // heat run
        report_performance(run());
        if (parser.get_check_flag()) {
            HostCSR h(graph);
            equal_components(components.to_host(), seq_tarjan(h));
        }
        components.reorder_labels_to_original();
        dump_array(parser.dump, components.to_host());
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
