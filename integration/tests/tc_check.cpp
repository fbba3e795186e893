// tc_check -- a program against the VGL tree WITH the HIP binding applied (oracle/Makefile, target `binding`; not part of the reference).
//
// TransitiveClosure::vgl_purdoms (algorithms/tc/tc.hpp:5-163) UNCHANGED on the HIP backend: SCC labels, the edge filter through an EdgesArray indexed by
// global_edge_pos, ParallelPrimitives::copy_if_indexes on a device condition (vgl_compute_api/hip/parallel_primitives_hip.h), reduce(REDUCE_MAX), the
// condensed graph as an EDGES_LIST_GRAPH and BFS::fast_vgl_top_down over it (the class's edges-list workers).  The reference's own `tc -check` compares
// zero elements (apps/tc/tc.cpp:67: verify_results(..., 0)) and its checker reads levels[i] where it means levels[end_vertex] (tc.hpp:183), so the
// answers are checked HERE: every pair against a sequential breadth-first search over the same graph through VGL_Graph's public accessors.
//
// usage: vgl_hip_tc_check -s <scale> -e <edge factor> -type rmat|ru -format csr [-it <pairs>]          prints TC CHECK PASSED / FAILED

#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE VECTOR_LENGTH*MAX_SX_AURORA_THREADS*128
#define VECTOR_CORE_THRESHOLD_VALUE 3*VECTOR_LENGTH

#include "graph_library.h"
#include <queue>

// is `to` reachable from `from` along stored edges (a vertex reaches itself)
static bool reachable(VGL_Graph &graph, int from, int to, std::vector<int> &seen, int stamp)
{
    if (from == to) return true;
    std::queue<int> todo;
    todo.push(from);
    seen[from] = stamp;
    while (!todo.empty()) {
        const int v = todo.front();
        todo.pop();
        const int n = graph.get_connections_count(v, SCATTER);
        for (int k = 0; k < n; k++) {
            const int u = graph.get_edge_dst(v, k, SCATTER);
            if (u == to) return true;
            if (seen[u] != stamp) { seen[u] = stamp; todo.push(u); }
        }
    }
    return false;
}

int main(int argc, char **argv)
{
    int failures = 0;
    try
    {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(VGL_RUNTIME::select_graph_format(parser), VGL_RUNTIME::select_graph_optimizations(parser));
        VGL_RUNTIME::prepare_graph(graph, parser);
        if (graph.get_container_type() != CSR_GRAPH) throw "tc_check: -format csr (pairs and answers are vertex ids of ONE numbering there)";
        const int vertices_count = graph.get_vertices_count();

        // pairs: random ones (mostly different components), plus pairs along stored edges and their reversals (same component or one-way)
        const int wanted = std::max(8, parser.get_number_of_rounds());
        vector<pair<int, int>> pairs;
        srand(20260105);
        for (int i = 0; i < wanted; i++) pairs.push_back(make_pair(rand() % vertices_count, rand() % vertices_count));
        for (int i = 0; i < wanted; i++) {
            const int v = rand() % vertices_count, n = graph.get_connections_count(v, SCATTER);
            if (n == 0) continue;
            const int u = graph.get_edge_dst(v, rand() % n, SCATTER);
            pairs.push_back(make_pair(v, u));
            pairs.push_back(make_pair(u, v));
        }
        vector<int> answer(pairs.size(), -1);
        TC::vgl_purdoms(graph, pairs, answer);

        std::vector<int> seen((size_t)vertices_count, 0);
        int yes = 0;
        for (size_t i = 0; i < pairs.size(); i++) {
            const bool truth = reachable(graph, pairs[i].first, pairs[i].second, seen, (int)i + 1);
            yes += truth;
            if ((answer[i] != 0) != truth) {
                failures++;
                if (failures <= 10) cout << "pair " << i << " (" << pairs[i].first << " -> " << pairs[i].second << "): vgl_purdoms says " << answer[i] << ", a sequential search says " << truth << endl;
            }
        }
        cout << pairs.size() << " pairs, " << yes << " reachable, " << failures << " wrong answers" << endl;
        if (yes == 0 || yes == (int)pairs.size()) { cout << "(the pairs do not tell the two answers apart)" << endl; failures++; }
        cout << (failures ? "TC CHECK FAILED" : "TC CHECK PASSED") << endl;
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 2; }
    catch (const char *error) { cout << error << endl; return 2; }
    return failures ? 1 : 0;
}
