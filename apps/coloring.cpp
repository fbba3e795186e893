// coloring app: counterpart of apps/coloring/coloring.cpp:10-44 (undirected input; -check = verify_colors: no edge joins two
// vertices of the same colour).  Runs entirely on the generic operator path: sparse frontiers, vertex post-ops, 64-bit vertex
// arrays, reduce<int>, generate_new_frontier.
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/coloring/coloring.cpp:3)
#include "common.hpp"
#include "algorithms/coloring.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, UNDIRECTED_GRAPH);
        VerticesArray<int> colors(graph);
        report_performance(Coloring::vgl_coloring(graph, colors));
        if (parser.get_check_flag()) {
            HostCSR h(graph);
            const std::vector<int> c = colors.to_host();
            long long errors = 0;
            int max_color = -1;
            for (int u = 0; u < h.V; u++) {
                if (c[u] < 0) errors++;
                max_color = std::max(max_color, c[u]);
                for (long long p = h.rowptr[u]; p < h.rowptr[u + 1]; p++)
                    if (h.adj[p] != u && c[h.adj[p]] == c[u]) errors++;
            }
            std::cout << "colors used: " << max_color + 1 << std::endl;
            std::cout << "error count: " << errors << std::endl;
        }
        colors.reorder(ORIGINAL);
        dump_array(parser.dump, colors.to_host());
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
