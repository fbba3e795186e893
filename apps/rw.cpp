// rw app: counterpart of apps/rw/rw.cpp:8-60 (-wv = percentage of walk vertices, -it = walk length, undirected generated input).
// -dump: int32 walk_results[V] indexed by ORIGINAL id, values ORIGINAL ids (DEAD_END = -1 for vertices that do not walk or got stuck).
// -check re-walks every walk on the host with the same draws and compares (possible because the draws are counter based).
#include "common.hpp"
#include "algorithms/rw.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, UNDIRECTED_GRAPH);
        const int V = graph.get_vertices_count();
        const int walk_length = parser.get_number_of_rounds();
        VerticesArray<int> walk_results(graph);
        const double msteps = RW::vgl_random_walk(graph, parser.walk_vertices_percent, walk_length, parser.seed, walk_results);
        std::cout << "AVG_PERF: " << msteps << " M walk steps/s" << std::endl;
        const std::vector<int> stored = walk_results.to_host();
        std::vector<int> original((size_t)V);
        for (int s = 0; s < V; s++) original[(size_t)graph.reorder(s, SCATTER, ORIGINAL)] = stored[(size_t)s] == DEAD_END ? DEAD_END : graph.reorder(stored[(size_t)s], SCATTER, ORIGINAL);
        if (parser.get_check_flag()) {
            HostCSR h(graph);
            std::vector<int> check((size_t)V, DEAD_END);
            for (int s = 0; s < V; s++) {
                const int walk = graph.reorder(s, SCATTER, ORIGINAL);
                if (!RW::is_walk_vertex(parser.seed, walk, parser.walk_vertices_percent)) continue;
                int cur = s;
                for (int it = 0; it < walk_length && cur != DEAD_END; it++) {
                    const long long deg = h.rowptr[(size_t)cur + 1] - h.rowptr[(size_t)cur];
                    cur = deg > 0 ? h.adj[(size_t)(h.rowptr[(size_t)cur] + (long long)(RW::draw(parser.seed, (unsigned long long)it, (unsigned long long)walk) % (unsigned long long)deg))] : DEAD_END;
                }
                check[(size_t)s] = cur;
            }
            verify_results(stored, check);
        }
        dump_array(parser.dump, original);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
