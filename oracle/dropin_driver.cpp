// dropin_driver.cpp -- RUNNABLE drop-in proof (test infrastructure; the driver is written here, the algorithm code is the reference's).
// The reference's own algorithm headers -- algorithms/bfs/bfs.hpp, algorithms/sswp/widest_paths.h, algorithms/hits/hits.h, algorithms/scc/scc.h -- are
// included UNCHANGED from where they lie (-I /root/reference), instantiated on this repository's operator class (VGL_GRAPH_ABSTRACTIONS =
// GraphAbstractionsHIP, vectorgraphlibrary_amd/hip/vgl_hip.hpp) and executed on the GPU.  Built by `make -C oracle dropin` into
// oracle/_ref/dropin_hip (git-ignored; travels to the GPU box like the other built files); tests/test_dropin_run_gpu.py runs it and
// compares its dumps with the CPU oracle.  What precedes the includes is what the reference's umbrella header graph_library.h would
// have put in scope (tests/test_dropin_compile.py does the same for the compile-only proof).
#include "../vectorgraphlibrary_amd/hip/vgl_hip.hpp"
#include <map>
#include <queue>
#include <set>
#include <stack>
using namespace std;                 // the reference's umbrella header does this globally
#define UNVISITED_VERTEX -1          // algorithms/bfs/change_state/change_state.h:21-23
#define FIRST_LEVEL_VERTEX 1
class BFS {                          // declaration of algorithms/bfs/bfs.h:31-46 (its other includes are the NEC-only direction-optimising pieces)
public:
    template <typename _T> static void fast_vgl_top_down(VGL_Graph &_graph, VerticesArray<_T> &_levels, int _source_vertex,
                                                         VGL_GRAPH_ABSTRACTIONS &_graph_API, VGL_FRONTIER &_frontier);
    template <typename _T> static double vgl_top_down(VGL_Graph &_graph, VerticesArray<_T> &_levels, int _source_vertex);
};
#include "algorithms/bfs/bfs.hpp"
#include "algorithms/sswp/widest_paths.h"
#include "algorithms/hits/hits.h"
#include "algorithms/scc/scc.h"

template <class T> static void dump(const char *path, const std::vector<T> &a, bool append = false)
{
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) throw "cannot open the dump file";
    fwrite(a.data(), sizeof(T), a.size(), f);
    fclose(f);
}

// usage: dropin_hip <bfs|sswp|hits|scc> <rmat|ru> <scale> <edge factor> <seed> <source (original id) | steps> <csr|vcsr> <dump file>
int main(int argc, char **argv)
{
    try {
        if (argc != 9) throw "usage: dropin_hip <bfs|sswp|hits|scc> <rmat|ru> <scale> <ef> <seed> <source|steps> <csr|vcsr> <dump>";
        const std::string algo = argv[1], kind = argv[2], fmt = argv[7];
        const int scale = atoi(argv[3]), ef = atoi(argv[4]), arg = atoi(argv[6]);
        VGL_RUNTIME::init_library(argc, argv);
        GraphGenerationAPI::seed() = strtoull(argv[5], nullptr, 10);
        EdgesContainer ec;
        const int v = 1 << scale;
        if (kind == "rmat") GraphGenerationAPI::R_MAT(ec, v, (long long)v * ef, 57, 19, 19, 5);
        else GraphGenerationAPI::random_uniform(ec, v, (long long)v * ef);
        VGL_Graph graph(fmt == "vcsr" ? VECTOR_CSR_GRAPH : CSR_GRAPH);
        graph.import(ec);
        if (algo == "bfs") {
            VerticesArray<int> levels(graph, SCATTER);
            // arg >= 0: that ORIGINAL vertex; arg < 0: -arg rounds from deterministic random non-isolated sources (bench.py's operator_api leg)
            double perf = 0;
            const int rounds = arg < 0 ? -arg : 1;
            for (int i = 0; i < rounds; i++) {
                const int source = arg < 0 ? graph.select_random_nz_vertex(ORIGINAL, i) : arg;
                perf += BFS::vgl_top_down(graph, levels, graph.reorder(source, ORIGINAL, SCATTER)) / rounds;       // the reference's function
            }
            levels.reorder(ORIGINAL);
            dump(argv[8], levels.to_host());
            std::cout << "DROPIN bfs " << perf << " MTEPS" << std::endl;
        } else if (algo == "sswp") {
            EdgesArray<float> capacities(graph);
            capacities.set_all_random(MAX_WEIGHT);
            VerticesArray<float> widths(graph, SCATTER);
            const double perf = SSWP::vgl_dijkstra(graph, capacities, widths, arg);      // (widest_paths.hpp:13 converts the ORIGINAL id itself)
            widths.reorder(ORIGINAL);
            dump(argv[8], widths.to_host());
            std::cout << "DROPIN sswp " << perf << " MTEPS" << std::endl;
        } else if (algo == "hits") {
            VerticesArray<double> auth(graph, SCATTER), hub(graph, SCATTER);
            HITS::vgl_hits(graph, auth, hub, arg);
            auth.reorder(ORIGINAL); hub.reorder(ORIGINAL);
            dump(argv[8], auth.to_host());
            dump(argv[8], hub.to_host(), true);
            std::cout << "DROPIN hits" << std::endl;
        } else if (algo == "scc") {
            VerticesArray<int> components(graph, SCATTER);
            SCC::vgl_forward_backward(graph, components);       // the reference's trim + forward-backward (algorithms/scc/scc.hpp:268-300); labels are its tree ids
            components.reorder(ORIGINAL);
            dump(argv[8], components.to_host());
            std::cout << "DROPIN scc" << std::endl;
        } else throw "unknown algorithm";
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
