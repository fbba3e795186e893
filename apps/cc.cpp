// cc app: counterpart of apps/cc/cc.cpp:11-60 (undirected input, heat run + timed run).
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/cc/cc.cpp:3)
#include "common.hpp"
#include "algorithms/cc.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, UNDIRECTED_GRAPH);
        VerticesArray<int> components(graph, SCATTER);
        const bool symmetric = parser.compute_mode == Parser::GENERATE_NEW_GRAPH;      // generated UNDIRECTED_GRAPH inputs hold both directions of every edge
        auto run = [&]() { return parser.fused ? ConnectedComponents::hip_fused(graph, components, symmetric)
                                               : ConnectedComponents::vgl_shiloach_vishkin(graph, components, parser.declared); };
        run();                                   // heat run
        report_performance(run());
        if (parser.get_check_flag()) {
            HostCSR h(graph);
            equal_components(components.to_host(), seq_components(h));
        }
        components.reorder_labels_to_original();
        dump_array(parser.dump, components.to_host());
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
