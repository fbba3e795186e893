/*
 * vgl_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the VectorGraphLibrary (VGL) hot path as exercised by
 * algorithms/{bfs,sssp,pr,cc} on the vgl_compute_api/multicore backend, on plain
 * CSR storage (int64 row offsets, int32 column ids, identity vertex numbering).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libvgl_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  Every algorithm below is checked against the genuine
 * reference multicore build (oracle/_ref, built by `make -C oracle ref` from the
 * sources under /root/reference where they lie) by oracle/make_golden.py, whose
 * outputs are committed under tests/golden/ and re-checked by
 * tests/test_oracle_golden.py on every run.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference root).
 */
#ifndef VGL_ORACLE_H
#define VGL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- deterministic synthetic inputs (own spec, see DESIGN.md "Synthetic inputs");
 *      distributions follow vgl_runtime/graph_generation/graph_generation.hpp:5-51,94-187 ---- */
uint64_t vgo_splitmix64(uint64_t x);
/* pseudo-random bijection of [0, 2^scale): deterministic stand-in for
 * EdgesContainer::random_shuffle_edges (edges_container.h:215-233) */
uint32_t vgo_relabel(uint32_t v, int scale, uint64_t seed);
void vgo_gen_rmat(int scale, int64_t first_edge, int64_t count, uint64_t seed,
                  int a, int b, int c, int d, int relabel, int32_t *src, int32_t *dst);
void vgo_gen_uniform(int scale, int64_t first_edge, int64_t count, uint64_t seed,
                     int32_t *src, int32_t *dst);
/* f32 weights uniform in [0,100): one per INPUT edge (common_generator.hpp:23-36, settings.h:93) */
void vgo_gen_weights(int64_t first_edge, int64_t count, uint64_t seed, float *w);

/* ---- COO -> CSR, stable in input order (csr/import.hpp:3-68, edges_container.h:101-161,
 *      sorter.h:55-92: std::stable_sort of an iota index array by src) ---- */
void vgo_coo_to_csr(int32_t V, int64_t E, const int32_t *src, const int32_t *dst,
                    int64_t *rowptr, int32_t *adj, int64_t *perm /* E, may be NULL */);
/* VectCSR renumbering: sorted position = (degree desc, original id asc)
 * (vect_csr/import.hpp:61-99, sorter.h:64-68).  fwd[orig]=sorted, bwd[sorted]=orig */
void vgo_degree_renumber(int32_t V, const int64_t *rowptr, int32_t *fwd, int32_t *bwd);

/* ---- BFS (algorithms/bfs/bfs.hpp:6-51 + multicore/advance_worker.hpp:62-149 +
 *      multicore/generate_new_frontier.hpp:113-164; checker algorithms/bfs/seq_bfs.hpp:13-55) ---- */
typedef struct {
    int32_t levels;           /* number of BFS levels executed (frontiers expanded) */
    int64_t edges_examined;   /* sum of out-degrees of all frontier vertices (m_ex) */
    int64_t frontier_total;   /* sum over levels of |F_l| (n_front) */
    int64_t discovered;       /* vertices with level > 0 at the end, incl. source */
} vgo_bfs_stats;
void vgo_bfs_top_down(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t source,
                      int32_t *levels, vgo_bfs_stats *st, int parallel);
void vgo_bfs_seq(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t source, int32_t *levels);

/* ---- SSSP (algorithms/sssp/shortest_paths.hpp:85-163 push all-active;
 *      checker algorithms/sssp/seq_shortest_paths.hpp:9-68) ---- */
int32_t vgo_sssp_bellman_ford(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *w,
                              int32_t source, float *dist, int parallel); /* returns iterations */
/* ---- SCC (algorithms/scc/scc.hpp forward-backward; checker seq_tarjan, seq_scc.hpp): canonical labels = smallest id of the SCC ---- */
void vgo_scc_tarjan(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp);
/* ---- HITS (algorithms/hits/hits.hpp:5-100; checker seq_hits, hits.hpp:103-173), f64 ---- */
void vgo_hits(int32_t V, const int64_t *out_rowptr, const int32_t *out_adj, const int64_t *in_rowptr, const int32_t *in_adj,
              int32_t steps, double *auth, double *hub);
/* ---- SSWP (algorithms/sswp/widest_paths.hpp:5-76; checker seq_widest_paths.hpp:5-64) ---- */
int32_t vgo_sswp_bellman_ford(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *cap,
                              int32_t source, float *width, int parallel);
void vgo_sswp_seq(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *cap, int32_t source, float *width);
void vgo_sssp_dijkstra(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *w,
                       int32_t source, float *dist);

/* ---- PageRank (algorithms/pr/pr.hpp:7-149, checker algorithms/pr/seq_pr.hpp:6-114) ----
 * indeg_noloops[v] = #(u->v, u != v).  dangling_mode 0: f32 sequential sum (seq_pr.hpp:70-79),
 * 1: f64 sum rounded to f32 once (deterministic midpoint of the multicore OpenMP reduction). */
void vgo_indegree_noloops(int32_t V, int64_t E, const int64_t *rowptr, const int32_t *adj, int32_t *indeg);
void vgo_pagerank(int32_t V, const int64_t *rowptr, const int32_t *adj, const int32_t *indeg_noloops,
                  int iterations, int dangling_mode, float *ranks, int parallel);

/* ---- CC Shiloach-Vishkin (algorithms/cc/shiloach_vishkin.hpp:7-88;
 *      checker algorithms/cc/seq_bfs_based.hpp:6-56) ---- */
int32_t vgo_cc_sv(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp, int parallel); /* returns hook passes */
void vgo_cc_seq_bfs(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp);

/* FNV-1a 64 over raw bytes (fixture hashing) */
uint64_t vgo_fnv1a64(const void *data, int64_t nbytes);
int vgo_max_threads(void);
void vgo_set_threads(int n);   /* OpenMP team size of the parallel=1 runs (the GPU box grants fewer CPUs than it shows) */
#ifdef __cplusplus
}
#endif
#endif
