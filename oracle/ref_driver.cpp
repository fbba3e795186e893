/*
 * ref_driver.cpp -- ORACLE TOOLING (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A driver written for this repo that runs the GENUINE reference implementation
 * (vgl_compute_api/multicore, header-only) on a deterministic input file and dumps
 * the result arrays in ORIGINAL vertex numbering.  It #includes the reference's
 * umbrella header from /root/reference where it lies (-I /root/reference); nothing
 * from the reference is copied into this repository.  Built only by
 * `make -C oracle ref` into oracle/_ref/ (git-ignored) when /root/reference exists.
 *
 * One TU, compiled seven times (-DAPP_BFS / -DAPP_SSSP / -DAPP_PR / -DAPP_CC / -DAPP_SSWP / -DAPP_HITS / -DAPP_SCC) because
 * the reference's degree-range thresholds are compile-time macros set per app
 * (apps/bfs/bfs.cpp:3-7, apps/sssp/sssp.cpp:3-12, apps/pr/pr.cpp:3-5, apps/cc/cc.cpp:3-5, apps/sswp/sswp.cpp:3-5).
 *
 * usage: ref_driver_<app> <graph.el_container> <csr|vcsr> <out.bin> [app args]
 *   bfs : <source_original_id>                      -> int32 levels[V]
 *   sssp: <source_original_id> <weights.f32> <push|pull>  (csr only) -> f32 dist[V]
 *   pr  : <iterations>                              -> f32 ranks[V] (vgl) then f32 ranks[V] (seq)
 *   cc  : (none)                                    -> int32 comp[V] (vgl SV) then int32 comp[V] (seq bfs)
 *   sswp: <source_original_id> <capacities.f32>     (csr only) -> f32 width[V] (vgl) then f32 width[V] (seq)
 *   hits: <steps>                                   -> f64 auth[V], hub[V] (vgl) then f64 auth[V], hub[V] (seq)
 *   scc : (none)                                    -> int32 comp[V] (vgl forward-backward) then int32 comp[V] (seq Tarjan)
 */
#if defined(APP_BFS)
#define INT_ELEMENTS_PER_EDGE 4.0
#define NEC_VECTOR_ENGINE_THRESHOLD_VALUE  VECTOR_LENGTH * MAX_SX_AURORA_THREADS * 128
#define VECTOR_CORE_THRESHOLD_VALUE 2*VECTOR_LENGTH
#define COLLECTIVE_FRONTIER_TYPE_CHANGE_THRESHOLD 0.35
#elif defined(APP_SSSP)
#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE VECTOR_LENGTH*MAX_SX_AURORA_THREADS*128
#define VECTOR_CORE_THRESHOLD_VALUE 5*VECTOR_LENGTH
#elif defined(APP_PR)
#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE 2147483646
#define VECTOR_CORE_THRESHOLD_VALUE 5*VECTOR_LENGTH
#elif defined(APP_CC)
#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE VECTOR_LENGTH*MAX_SX_AURORA_THREADS*128
#define VECTOR_CORE_THRESHOLD_VALUE 5*VECTOR_LENGTH
#elif defined(APP_SSWP)
#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE VECTOR_LENGTH*MAX_SX_AURORA_THREADS*128
#define VECTOR_CORE_THRESHOLD_VALUE 3*VECTOR_LENGTH
#elif defined(APP_HITS)
#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE 2147483646
#define VECTOR_CORE_THRESHOLD_VALUE 5*VECTOR_LENGTH
#elif defined(APP_SCC)
#define INT_ELEMENTS_PER_EDGE 4.0
#define NEC_VECTOR_ENGINE_THRESHOLD_VALUE  VECTOR_LENGTH * MAX_SX_AURORA_THREADS * 128
#define VECTOR_CORE_THRESHOLD_VALUE VECTOR_LENGTH
#define COLLECTIVE_FRONTIER_TYPE_CHANGE_THRESHOLD 0.35
#else
#error "define one of APP_BFS / APP_SSSP / APP_PR / APP_CC / APP_SSWP / APP_HITS / APP_SCC"
#endif

#include "graph_library.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <typename T>
static void dump(FILE *f, VerticesArray<T> &a)
{
    a.reorder(ORIGINAL);
    fwrite(a.get_ptr(), sizeof(T), a.size(), f);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s graph.el_container csr|vcsr out.bin [args]\n", argv[0]); return 2; }
    try
    {
        VGL_RUNTIME::init_library(argc, argv);
        GraphStorageFormat fmt = (strcmp(argv[2], "vcsr") == 0) ? VECTOR_CSR_GRAPH : CSR_GRAPH;
        EdgesContainer ec;
        if (!ec.load_from_binary_file(argv[1])) { fprintf(stderr, "cannot read %s\n", argv[1]); return 3; }
        VGL_Graph graph(fmt);
        graph.import(ec);
        FILE *out = fopen(argv[3], "wb");
        if (!out) { fprintf(stderr, "cannot write %s\n", argv[3]); return 3; }

#if defined(APP_BFS)
        int source = graph.reorder(atoi(argv[4]), ORIGINAL, SCATTER);
        VerticesArray<int> levels(graph, SCATTER);
        BFS::vgl_top_down(graph, levels, source);
        dump(out, levels);
#elif defined(APP_SSSP)
        if (fmt != CSR_GRAPH) { fprintf(stderr, "sssp driver: csr only\n"); return 2; }
        int source_orig = atoi(argv[4]);
        bool pull = (argc > 6 && strcmp(argv[6], "pull") == 0);
        long long E = graph.get_edges_count();
        std::vector<float> w(E);
        FILE *wf = fopen(argv[5], "rb");
        if (!wf || fread(w.data(), sizeof(float), E, wf) != (size_t)E) { fprintf(stderr, "bad weights file\n"); return 3; }
        fclose(wf);
        EdgesArray<float> weights(graph);
        float *wp = weights.get_ptr();              /* CSR layout [out E ; in E] */
        for (long long p = 0; p < E; p++) wp[p] = w[p];
        graph.copy_outgoing_to_incoming_edges(wp, wp + E);
        TraversalDirection dir = pull ? GATHER : SCATTER;
        VerticesArray<float> dist(graph, dir);
        int source = graph.reorder(source_orig, ORIGINAL, dir);
        ShortestPaths::vgl_dijkstra(graph, weights, dist, source, ALL_ACTIVE, pull ? PULL_TRAVERSAL : PUSH_TRAVERSAL);
        dump(out, dist);
#elif defined(APP_PR)
        int iters = atoi(argv[4]);
        VerticesArray<float> ranks(graph);
        PageRank::vgl_page_rank(graph, ranks, 1.0e-4f, iters);
        dump(out, ranks);
        VerticesArray<float> seq_ranks(graph);
        PageRank::seq_page_rank(graph, seq_ranks, 1.0e-4f, iters);
        dump(out, seq_ranks);
#elif defined(APP_CC)
        VerticesArray<int> comp(graph, SCATTER);
        ConnectedComponents::vgl_shiloach_vishkin(graph, comp);
        dump(out, comp);
        VerticesArray<int> check(graph, SCATTER);
        ConnectedComponents::seq_bfs_based(graph, check);
        dump(out, check);
#elif defined(APP_SSWP)
        if (fmt != CSR_GRAPH) { fprintf(stderr, "sswp driver: csr only\n"); return 2; }
        int source_orig = atoi(argv[4]);
        long long E = graph.get_edges_count();
        std::vector<float> w(E);
        FILE *wf = fopen(argv[5], "rb");
        if (!wf || fread(w.data(), sizeof(float), E, wf) != (size_t)E) { fprintf(stderr, "bad capacities file\n"); return 3; }
        fclose(wf);
        EdgesArray<float> capacities(graph);
        float *wp = capacities.get_ptr();           /* CSR layout [out E ; in E] */
        for (long long p = 0; p < E; p++) wp[p] = w[p];
        graph.copy_outgoing_to_incoming_edges(wp, wp + E);
        VerticesArray<float> widths(graph, SCATTER);
        SSWP::vgl_dijkstra(graph, capacities, widths, source_orig);      /* takes the ORIGINAL id (widest_paths.hpp:13) */
        dump(out, widths);
        VerticesArray<float> check(graph, SCATTER);
        SSWP::seq_dijkstra(graph, capacities, check, source_orig);
        dump(out, check);
#elif defined(APP_HITS)
        int steps = atoi(argv[4]);
        VerticesArray<double> auth(graph), hub(graph);
        HITS::vgl_hits(graph, auth, hub, steps);
        dump(out, auth);
        dump(out, hub);
        VerticesArray<double> check_auth(graph), check_hub(graph);
        HITS::seq_hits(graph, check_auth, check_hub, steps);
        dump(out, check_auth);
        dump(out, check_hub);
#elif defined(APP_SCC)
        VerticesArray<int> comp(graph, SCATTER);
        SCC::vgl_forward_backward(graph, comp);
        dump(out, comp);
        VerticesArray<int> check(graph, SCATTER);
        SCC::seq_tarjan(graph, check);
        dump(out, check);
#endif
        fclose(out);
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 1; }
    catch (const char *error) { cout << error << endl; return 1; }
    return 0;
}
