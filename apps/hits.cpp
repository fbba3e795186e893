// hits app: counterpart of apps/hits/hits.cpp:17-62 (f64; -it = number of steps; -check against a sequential run).
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/hits/hits.cpp:3)
#include "common.hpp"
#include "algorithms/hits.hpp"
#define base_type double
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        VerticesArray<base_type> auth(graph), hub(graph);
        const int steps = parser.get_number_of_rounds();
        report_performance(parser.fused ? HITS::hip_fused(graph, auth, hub, steps) : HITS::vgl_hits(graph, auth, hub, steps));
        std::vector<base_type> a = auth.to_host(), h = hub.to_host();
        if (parser.get_check_flag()) {
            HostCSR out(graph, SCATTER), in(graph, GATHER);
            std::vector<double> ra, rh;
            seq_hits(out, in, steps, ra, rh);
            double worst = 0;                                      // largest relative deviation from the sequential run
            for (size_t i = 0; i < ra.size(); i++) {
                worst = std::max(worst, std::fabs(a[i] - ra[i]) / std::max(std::fabs(ra[i]), 1e-300));
                worst = std::max(worst, std::fabs(h[i] - rh[i]) / std::max(std::fabs(rh[i]), 1e-300));
            }
            std::cout << "error count: " << (worst <= 1e-9 ? 0 : 1) << std::endl;
        }
        auth.reorder(ORIGINAL); hub.reorder(ORIGINAL);
        a = auth.to_host(); h = hub.to_host();
        a.insert(a.end(), h.begin(), h.end());                     // dump: authorities then hubs (ORIGINAL numbering)
        dump_array(parser.dump, a);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
