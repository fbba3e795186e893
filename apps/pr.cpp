// pr app: counterpart of apps/pr/pr.cpp:11-63 (-it = number of iterations).
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/pr/pr.cpp:3)
#include "common.hpp"
#include "algorithms/pr.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        VerticesArray<float> page_ranks(graph);
        const double perf = parser.fused ? PageRank::hip_fused(graph, page_ranks, parser.get_number_of_rounds())
                                         : PageRank::vgl_page_rank(graph, page_ranks, 1.0e-4f, parser.get_number_of_rounds(),
                                                                   parser.declared ? 3 : parser.traversal == Parser::PULL_TRAVERSAL ? 2 : parser.deterministic ? 1 : 0);
        report_performance(perf);
        if (parser.get_check_flag()) {
            HostCSR h(graph);
            const std::vector<float> ref = seq_page_rank(h, parser.get_number_of_rounds()), got = page_ranks.to_host();
            double diff = 0;                                          // verify_ranking_results (verify_results.h:115-130)
            for (size_t i = 0; i < ref.size(); i++) diff += std::fabs((double)ref[i] - got[i]);
            std::cout << "error count: " << (diff / ref.size() < 1e-4 ? 0 : 1) << std::endl;
        }
        page_ranks.reorder(ORIGINAL);
        dump_array(parser.dump, page_ranks.to_host());
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
