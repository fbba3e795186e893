// tc app: counterpart of apps/tc/tc.cpp:8-70 (-it = number of vertex pairs; -check compares Purdom's algorithm with one BFS per
// source).  -dump writes int32 triples (first, second, answer) with ORIGINAL vertex ids.
#define INT_ELEMENTS_PER_EDGE 5.0      // VGL byte accounting of this app (apps/tc/tc.cpp:3)
#include "common.hpp"
#include "algorithms/tc.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser, DIRECTED_GRAPH);
        const int V = graph.get_vertices_count();
        const int pairs_count = std::max(1, parser.get_number_of_rounds());
        TC::Pairs original_pairs, vertex_pairs;                      // deterministic stand-in for rand() % V (tc.cpp:13)
        unsigned long long x = 0x9E3779B97F4A7C15ULL ^ parser.seed;
        auto next = [&x]() { x += 0x9E3779B97F4A7C15ULL; unsigned long long z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
        for (int i = 0; i < pairs_count; i++) {
            const int a = (int)(next() % (unsigned long long)V), b = (int)(next() % (unsigned long long)V);
            original_pairs.push_back({a, b});
            vertex_pairs.push_back({graph.reorder(a, ORIGINAL, SCATTER), graph.reorder(b, ORIGINAL, SCATTER)});
        }
        std::vector<int> answer(vertex_pairs.size(), 0);
        report_performance(parser.bfs_based ? TC::vgl_bfs_based(graph, vertex_pairs, answer) : TC::vgl_purdoms(graph, vertex_pairs, answer));
        if (parser.get_check_flag()) {
            std::vector<int> check_answer(vertex_pairs.size(), 0);
            if (parser.bfs_based) TC::vgl_purdoms(graph, vertex_pairs, check_answer); else TC::vgl_bfs_based(graph, vertex_pairs, check_answer);
            verify_results(answer, check_answer);
        }
        std::vector<int> triples;
        for (size_t i = 0; i < answer.size(); i++) { triples.push_back(original_pairs[i].first); triples.push_back(original_pairs[i].second); triples.push_back(answer[i]); }
        dump_array(parser.dump, triples);
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
