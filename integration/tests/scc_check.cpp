// scc_check -- a program against the VGL tree WITH the HIP binding applied (oracle/Makefile, target `binding`; not part of the reference).
//
// SCC::vgl_forward_backward (algorithms/scc/scc.hpp, unchanged) on SEEDED graphs, so that a run whose components differ from SCC::seq_tarjan can be
// repeated: the reference's apps seed their generators with time(NULL).  For every seed in [first, first + count): an R-MAT-like edge list from a
// splitmix64 stream (same quadrant probabilities as graph_generation.hpp:131-169), VGL_Graph::import, the algorithm, the sequential checker, and the
// number of vertices whose component disagrees (the rule of equal_components, verify_results.h:198-246).  A seed that fails is run again `repeat`
// times: the same count every time means the graph decides, a varying one means a race.
//
// usage: vgl_hip_scc_check <scale> <edge factor> <first seed> <count> [repeat] [csr|vcsr]          prints SCC CHECK PASSED / FAILED

#define INT_ELEMENTS_PER_EDGE 4.0
#define NEC_VECTOR_ENGINE_THRESHOLD_VALUE  VECTOR_LENGTH * MAX_SX_AURORA_THREADS * 128
#define VECTOR_CORE_THRESHOLD_VALUE VECTOR_LENGTH
#define COLLECTIVE_FRONTIER_TYPE_CHANGE_THRESHOLD 0.35

#include "graph_library.h"
#include <map>

static inline unsigned long long splitmix64(unsigned long long &state)
{
    unsigned long long z = (state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void fill_rmat(EdgesContainer &edges, int scale, int edge_factor, unsigned long long seed)
{
    const int vertices_count = 1 << scale;
    const long long edges_count = (long long)vertices_count * edge_factor;
    edges.resize(vertices_count, edges_count);
    int *src_ids = edges.get_src_ids(), *dst_ids = edges.get_dst_ids();
    unsigned long long state = seed * 0x2545F4914F6CDD1DULL + 12345;
    for (long long e = 0; e < edges_count; e++) {
        int row = 0, col = 0;
        for (int bit = 0; bit < scale; bit++) {
            const unsigned r = (unsigned)(splitmix64(state) % 100);
            const int quadrant = r < 57 ? 0 : (r < 76 ? 1 : (r < 95 ? 2 : 3));
            row = (row << 1) | (quadrant >> 1);
            col = (col << 1) | (quadrant & 1);
        }
        // a fixed scramble of the ids, so that hubs are not the first vertices (the reference relabels randomly, edges_container.h:215-233)
        src_ids[e] = (int)(((unsigned long long)row * 2654435761ULL + 7) % (unsigned long long)vertices_count);
        dst_ids[e] = (int)(((unsigned long long)col * 2654435761ULL + 7) % (unsigned long long)vertices_count);
    }
}

static int disagreements(VerticesArray<int> &first, VerticesArray<int> &second)
{
    first.reorder(ORIGINAL);
    second.reorder(ORIGINAL);
    std::map<int, int> f_s, s_f;
    const int vertices_count = first.size();
    for (int i = 0; i < vertices_count; i++) { f_s[first[i]] = second[i]; s_f[second[i]] = first[i]; }
    int wrong = 0;
    for (int i = 0; i < vertices_count; i++) wrong += (f_s[first[i]] != second[i]) || (s_f[second[i]] != first[i]);
    return wrong;
}

static int run_seed(int scale, int edge_factor, unsigned long long seed, GraphStorageFormat format)
{
    EdgesContainer edges;
    fill_rmat(edges, scale, edge_factor, seed);
    VGL_Graph graph(format);
    graph.import(edges);
    VerticesArray<int> components(graph, SCATTER), check_components(graph, SCATTER);
    SCC::vgl_forward_backward(graph, components);
    SCC::seq_tarjan(graph, check_components);
    return disagreements(components, check_components);
}

int main(int argc, char **argv)
{
    int failures = 0;
    try
    {
        VGL_RUNTIME::init_library(argc, argv);
        if (argc < 5) throw "usage: vgl_hip_scc_check <scale> <edge factor> <first seed> <count> [repeat] [csr|vcsr]";
        const int scale = atoi(argv[1]), edge_factor = atoi(argv[2]), count = atoi(argv[4]), repeat = argc > 5 ? atoi(argv[5]) : 3;
        const unsigned long long first = strtoull(argv[3], NULL, 10);
        const GraphStorageFormat format = (argc > 6 && string(argv[6]) == "vcsr") ? VECTOR_CSR_GRAPH : CSR_GRAPH;
        streambuf *console = cout.rdbuf();
        ostringstream quiet;
        for (unsigned long long seed = first; seed < first + (unsigned long long)count; seed++) {
            cout.rdbuf(quiet.rdbuf());                          // (the algorithm prints its trim steps and component sizes)
            const int wrong = run_seed(scale, edge_factor, seed, format);
            cout.rdbuf(console);
            quiet.str("");
            if (wrong == 0) continue;
            failures++;
            cout << "seed " << seed << ": " << wrong << " vertices disagree; again:";
            for (int k = 0; k < repeat; k++) {
                cout.rdbuf(quiet.rdbuf());
                const int again = run_seed(scale, edge_factor, seed, format);
                cout.rdbuf(console);
                quiet.str("");
                cout << " " << again;
            }
            cout << endl;
        }
        cout << count << " seeds, " << failures << " with components that differ from seq_tarjan" << endl;
        cout << (failures ? "SCC CHECK FAILED" : "SCC CHECK PASSED") << endl;
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 2; }
    catch (const char *error) { cout << error << endl; return 2; }
    return failures ? 1 : 0;
}
