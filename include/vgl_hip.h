/*
 * vgl_hip.h -- C ABI of libvgl_hip.so, the MI355X (gfx950) backend for VectorGraphLibrary's
 * frontier-driven advance / compute / reduce / generate_new_frontier hot path.
 *
 * Boundary: the reference selects a backend at compile time as a C++ template class
 * (architecture_independent_api.h:33-43; member list vgl_compute_api/template/
 * graph_abstractions_template.h:44-104; recipe manuals/add_new_architecture.txt:1-7).
 * This C ABI is the layer that class binds to (see INTEGRATION.md and
 * vectorgraphlibrary_amd/hip/vgl_hip.hpp): plain pointers and sizes, opaque
 * handles, `int` status (0 = ok) + vgl_hip_last_error().  All device pointers are raw HIP
 * device pointers owned by the caller unless stated; every call is ordered on the context's
 * stream and returns after the work is ENQUEUED unless it has a host-visible result, in which
 * case it synchronises the stream (the reference GPU backend is synchronous per primitive:
 * vgl_compute_api/gpu/advance_csr.hpp:204).
 *
 * Vertex ids are int32, edge offsets int64, properties 4-byte.
 */
#ifndef VGL_HIP_H
#define VGL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VGL_HIP_ABI_VERSION 1

typedef struct vgl_hip_ctx vgl_hip_ctx;
typedef struct vgl_hip_graph vgl_hip_graph;
typedef struct vgl_hip_frontier vgl_hip_frontier;

/* ---- context / errors (replaces VGL_RUNTIME::init_library cudaSetDevice path, vgl_runtime.hpp:5-25,
 *      and the throw "literal" convention, apps/bfs/bfs.cpp:53-61) ---- */
int vgl_hip_abi_version(void);
const char *vgl_hip_last_error(void);
/* stream: a hipStream_t created by the caller (e.g. torch's current stream); NULL = the device's default stream */
int vgl_hip_ctx_create(int device, void *stream, vgl_hip_ctx **out);
int vgl_hip_ctx_destroy(vgl_hip_ctx *ctx);
int vgl_hip_ctx_sync(vgl_hip_ctx *ctx);
/* plan builders take their scratch and plan arrays from the device's stream-ordered memory pool, which keeps freed memory cached (a
 * second build of the same size then pays no allocator or first-touch cost); this hands the cached memory back to the driver */
int vgl_hip_ctx_trim(vgl_hip_ctx *ctx);
void *vgl_hip_ctx_stream(vgl_hip_ctx *ctx);

/* ---- device memory (replaces MemoryAPI::allocate_array / move_array_to_device, memory_API.hpp:4-15,100-110) ---- */
int vgl_hip_malloc(vgl_hip_ctx *ctx, size_t bytes, void **dptr);
int vgl_hip_free(vgl_hip_ctx *ctx, void *dptr);
int vgl_hip_memcpy_h2d(vgl_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int vgl_hip_memcpy_d2h(vgl_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int vgl_hip_memset(vgl_hip_ctx *ctx, void *dst, int byte_value, size_t bytes);

/* ---- synthetic inputs on the device (GraphGenerationAPI::R_MAT / random_uniform,
 *      graph_generation.hpp:5-51,94-187; weights common_generator.hpp:23-36).  Counter-based, so
 *      any [first_edge, first_edge+count) slice can be produced independently on any rank. ---- */
int vgl_hip_gen_rmat(vgl_hip_ctx *ctx, int scale, int64_t first_edge, int64_t count, uint64_t seed,
                     int a, int b, int c, int d, int relabel, int32_t *d_src, int32_t *d_dst);
int vgl_hip_gen_uniform(vgl_hip_ctx *ctx, int scale, int64_t first_edge, int64_t count, uint64_t seed,
                        int32_t *d_src, int32_t *d_dst);
int vgl_hip_gen_weights(vgl_hip_ctx *ctx, int64_t first_edge, int64_t count, uint64_t seed, float *d_w);

/* ---- COO -> CSR on the device, stable in input order (CSRGraph::import, csr/import.hpp:3-68).
 *      Only edges with row_begin <= src < row_end are kept (edge-cut shard, vect_csr/get_api.hpp:66-94);
 *      d_rowptr has (row_end-row_begin+1) entries rebased to 0, d_adj / d_perm have capacity `count`.
 *      d_perm (optional) receives the INPUT edge index of every CSR position (edges_reorder_indexes).
 *      *kept_out = number of edges kept.  Synchronises. ---- */
int vgl_hip_coo_to_csr(vgl_hip_ctx *ctx, int32_t V, int64_t count, const int32_t *d_src, const int32_t *d_dst,
                       int32_t row_begin, int32_t row_end,
                       int64_t *d_rowptr, int32_t *d_adj, int64_t *d_perm, int64_t *kept_out);
/* out[i] = in[perm[i]] for 4-byte elements (EdgesArray weights follow the CSR order,
 * csr_edges_array.hpp:31-40) */
int vgl_hip_gather_u32(vgl_hip_ctx *ctx, int64_t n, const int64_t *d_perm, const void *d_in, void *d_out);
/* VectCSR-style vertex renumbering (VectorCSRGraph::import, vect_csr/import.hpp:61-99): sorted position =
 * (degree descending, original id ascending).  degree_kind 0 = out-degree, 1 = in-degree, 2 = in+out.  The reference
 * renumbers each direction separately; this backend keeps ONE numbering for both directions so that top-down and
 * bottom-up steps share vertex arrays.  d_fwd[orig] = sorted id, d_bwd[sorted] = orig id (forward/backward_conversion).
 * Hot (high-degree) vertices become contiguous, so the per-edge 4-byte gathers of dist/levels/labels/contrib[dst]
 * mostly hit the XCD L2 instead of costing a fabric line each.  Synchronises. */
int vgl_hip_degree_order(vgl_hip_ctx *ctx, int32_t V, int64_t count, const int32_t *d_src, const int32_t *d_dst,
                         int degree_kind, int32_t *d_fwd, int32_t *d_bwd);
/* the two halves of vgl_hip_degree_order for inputs that are produced in chunks (scale-27 shards): accumulate degrees of a
 * chunk into d_degree (uint32[V], zeroed by the caller), then derive the order from the finished histogram. */
int vgl_hip_degree_hist_add(vgl_hip_ctx *ctx, int64_t count, const int32_t *d_src, const int32_t *d_dst, int degree_kind,
                            uint32_t *d_degree);
int vgl_hip_degree_order_from_degrees(vgl_hip_ctx *ctx, int32_t V, const uint32_t *d_degree, int32_t *d_fwd, int32_t *d_bwd);
/* out[i] = map[in[i]] (relabel ids) and out[i] = in[idx[i]] (permute a vertex array) for 4-byte elements */
int vgl_hip_relabel_i32(vgl_hip_ctx *ctx, int64_t n, const int32_t *d_map, const int32_t *d_in, int32_t *d_out);
int vgl_hip_permute_u32(vgl_hip_ctx *ctx, int64_t n, const int32_t *d_idx, const void *d_in, void *d_out);
/* CC labels computed on a renumbered graph -> labels in ORIGINAL ids: out[orig v] = min original id of v's component.
 * d_comp: labels in sorted numbering (int32[V]); d_scratch: int32[V]. */
int vgl_hip_cc_labels_to_original(vgl_hip_ctx *ctx, int32_t V, const int32_t *d_comp, const int32_t *d_fwd,
                                  const int32_t *d_bwd, int32_t *d_scratch, int32_t *d_out);

/* edge-balanced contiguous vertex ranges (VectorCSRGraph::get_mpi_thresholds, vect_csr/get_api.hpp:66-94):
 * bounds[p] .. bounds[p+1] own ~E/parts edges each.  Host array of parts+1 entries.  Synchronises. */
int vgl_hip_partition_rows(vgl_hip_ctx *ctx, int32_t V, const int64_t *d_rowptr, int parts, int32_t *bounds_host);

/* ---- graph handle: borrows the caller's device CSR (CSRGraph, csr/csr_graph.h:22-87; VGL_Graph holds an
 *      outgoing and an incoming container, vgl_graph.h:7-79).  Rows [row_begin,row_end) are present in each
 *      direction (whole graph: 0..V).  The incoming direction may be NULL when only push algorithms run.
 *      Creation builds the per-tile row tables the edge-balanced kernels use (derived data, owned). ---- */
int vgl_hip_graph_create(vgl_hip_ctx *ctx, int32_t V, int32_t row_begin, int32_t row_end,
                         const int64_t *d_out_rowptr, const int32_t *d_out_adj, int64_t out_edges,
                         const int64_t *d_in_rowptr, const int32_t *d_in_adj, int64_t in_edges,
                         vgl_hip_graph **out);
int vgl_hip_graph_destroy(vgl_hip_ctx *ctx, vgl_hip_graph *g);

/* ---- frontier (BaseFrontier, base_frontier.h:5-62; sparsity enum framework_types.h:156-160) ---- */
#define VGL_HIP_FRONTIER_DENSE 0
#define VGL_HIP_FRONTIER_SPARSE 1
#define VGL_HIP_FRONTIER_ALL_ACTIVE 2
int vgl_hip_frontier_create(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_frontier **out);
int vgl_hip_frontier_destroy(vgl_hip_ctx *ctx, vgl_hip_frontier *f);
/* a handle over flags / ids arrays the caller owns (device-accessible int32[V] each) -- for a backend bound to the reference's own frontier
 * containers (FrontierCSR / FrontierVectorCSR, base_frontier.h:5-62), whose host code reads and writes those arrays; set_state tells the handle
 * what they hold (size, sum of degrees, sparsity as VGL_HIP_FRONTIER_*) and which graph handle (direction container) the frontier refers to. */
int vgl_hip_frontier_create_on(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_flags, int32_t *d_ids, vgl_hip_frontier **out);
int vgl_hip_frontier_set_state(vgl_hip_ctx *ctx, vgl_hip_frontier *f, vgl_hip_graph *g, int32_t size, int64_t neighbours, int sparsity);
int vgl_hip_frontier_set_all_active(vgl_hip_ctx *ctx, vgl_hip_frontier *f);      /* frontier/.../modification.hpp set_all_active */
int vgl_hip_frontier_clear(vgl_hip_ctx *ctx, vgl_hip_frontier *f);
int vgl_hip_frontier_add_vertex(vgl_hip_ctx *ctx, vgl_hip_frontier *f, int32_t v); /* only into an empty frontier (modification.hpp:33-36) */
int vgl_hip_frontier_info(vgl_hip_ctx *ctx, vgl_hip_frontier *f, int32_t *size, int64_t *neighbours, int *sparsity);
int32_t *vgl_hip_frontier_ids(vgl_hip_frontier *f);    /* device, ascending ids, `size` valid entries when SPARSE */
int32_t *vgl_hip_frontier_flags(vgl_hip_frontier *f);  /* device, int32[V] 0/1 */
/* generate_new_frontier from a caller-filled flags array (generate_new_frontier_worker(CSRGraph&),
 * multicore/generate_new_frontier.hpp:113-164 + copy_if_indexes, copy_if.hpp:128-191): counts, Σdegree,
 * ALL_ACTIVE when size == V else SPARSE with ascending-id compaction.  d_flags may alias the frontier's own flags.
 * dense_threshold > 0 selects the VectCSR rule (size > threshold*V => DENSE, flags only; generate_new_frontier.hpp:67-91). */
int vgl_hip_gnf_from_flags(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *d_flags, double dense_threshold,
                           vgl_hip_frontier *f);
/* same with the predicate (d_values[v] == value) evaluated in-kernel (BFS on_next_level, bfs.hpp:40-45) */
int vgl_hip_gnf_equal_i32(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *d_values, int32_t value,
                          double dense_threshold, vgl_hip_frontier *f);

/* ---- plans for the templated advance kernels of the C++ operator class (vectorgraphlibrary_amd/hip/): the edge-balanced
 *      kernels need, per direction (0 = outgoing / SCATTER, 1 = incoming / GATHER), the static tile->row table of the
 *      graph and, for a SPARSE frontier, the exclusive edge offsets of its ids plus the tile->position table. ---- */
int vgl_hip_graph_tile_rows(vgl_hip_graph *g, int direction, const int32_t **d_tile_row, int64_t *ntiles);
int vgl_hip_frontier_advance_plan(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_frontier *f, int direction,
                                  const int64_t **d_offs, const int32_t **d_tile_first, int64_t *edges);
/* generate_new_frontier with a USER predicate (the C++ operator class).  The count pass -- predicate, flags, per-tile counts, totals --
 * runs in the caller's translation unit: vgl_k_gnf_count of csrc/vgl_gnf.h instantiated with the user's lambda, launched on the context's
 * stream with the buffers vgl_hip_gnf_begin hands out (plain pointers: the caller never sees the library's internal structures).
 * vgl_hip_gnf_complete waits for that launch (sequence number `seq` from begin, published by the kernel's last workgroup) and does the
 * rest: size / neighbours / sparsity choice (generate_new_frontier.hpp:67-91,113-164) and, for a SPARSE result, the ascending-id
 * compaction -- with the exclusive out-edge offsets of the ids when want_plan != 0, which vgl_hip_frontier_advance_plan(direction 0) reuses.
 * Round 5: the count pass writes the predicate's bits as a BITMAP (V / 8 bytes) instead of V int32 flags, the compaction reads the bitmap, and only a
 * DENSE / ALL_ACTIVE result gets its int32 flags (expanded from the bitmap): a BFS level on RMAT-24 moved 64 MiB of flags out and in again for a frontier
 * of a few thousand ids.  VGL_GNF_INT_FLAGS=1 keeps the flags of every result (the reference's contract to the letter). */
typedef struct {
    int32_t nrows, row_begin;          /* owned rows of the graph handle (generate_new_frontier needs a whole-graph handle) */
    int64_t nvtiles;                   /* 2048-vertex tiles = workgroups of the count launch (256 threads each) */
    const int64_t *out_rowptr;         /* degrees counted into the neighbour total */
    int32_t *vt_cnt, *vt_cnt_off;      /* per-tile counts and their exclusive offsets */
    int64_t *vt_deg, *vt_deg_off;
    uint32_t *ticket;                  /* arrival counters of the launch (the last workgroup scans the tiles) */
    int64_t *counters;                 /* device counter slots */
    volatile int64_t *host_counters;   /* their pinned mirror */
    int32_t *flags;                    /* the frontier's int32 flags for the count pass to write, or NULL (round 5, the default): the pass then leaves only
                                          front_bytes and vgl_hip_gnf_complete writes the flags of a DENSE / ALL_ACTIVE result itself -- the flags of a
                                          SPARSE frontier are not materialised (nothing reads them: every primitive walks its ids) */
    uint8_t *front_bytes;              /* the predicate's bits, byte v >> 3, bit v & 7 (always written by the count pass) */
    int64_t *plan_offs;                /* want_plan: edge-offset array whose terminator the count pass writes, else NULL */
    int64_t seq;                       /* sequence number the launch must publish */
} vgl_hip_gnf_buffers;
int vgl_hip_gnf_begin(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_frontier *f, int want_plan, vgl_hip_gnf_buffers *out);
int vgl_hip_gnf_complete(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_frontier *f, double dense_threshold, int want_plan, int64_t seq);
/* sums n doubles on the device in a fixed order (the operator class folds its per-workgroup reduce partials with it) */
int vgl_hip_reduce_sum_f64_buffer(vgl_hip_ctx *ctx, int64_t n, const double *d_values, double *result);

/* ---- reduce (reduce_worker_sum, multicore/reduce.hpp:6-60; only REDUCE_SUM is live).
 *      Sums d_values[v] over the frontier's active vertices; deterministic (fixed tree). Synchronises. ---- */
int vgl_hip_reduce_sum_i32(vgl_hip_ctx *ctx, vgl_hip_frontier *f, const int32_t *d_values, int64_t *result);
int vgl_hip_reduce_sum_f32(vgl_hip_ctx *ctx, vgl_hip_frontier *f, const float *d_values, double *result);
/* number of v with a[v] != b[v] (SSSP reduce_changes, shortest_paths.hpp:143-152) */
int vgl_hip_count_not_equal_u32(vgl_hip_ctx *ctx, int32_t n, const void *d_a, const void *d_b, int64_t *result);

/* ---- fused algorithm fast paths (operators of algorithms/{bfs,sssp,pr,cc}) ---- */
typedef struct {
    int32_t levels;            /* frontiers expanded */
    int32_t td_steps, bu_steps;
    int64_t edges_examined;    /* adjacency entries actually read by advance kernels */
    int64_t frontier_total;    /* sum of |F_l| */
    int64_t discovered;        /* vertices reached, incl. source */
    int64_t algorithmic_bytes; /* 8*m_ex + 20*n_front + 4*n_disc + 4*V (+ V/8 per bottom-up level) */
    int64_t td_edges, td_frontier;   /* per-kernel shares for the roofline line: top-down launches */
    int64_t bu_edges, bu_found;      /* bottom-up launches: adjacency entries probed, vertices discovered */
} vgl_hip_bfs_stats;
#define VGL_HIP_BFS_TOP_DOWN 0           /* BFS::fast_vgl_top_down, bfs.hpp:6-51 */
#define VGL_HIP_BFS_DIRECTION_OPT 1      /* + bottom-up steps; switch rule change_state.hpp:100-141 (ALPHA 15, BETA 18) */
int vgl_hip_bfs_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t source, int mode,
                    int32_t *d_levels, vgl_hip_bfs_stats *stats);
/* `count` traversals from HOST source ids one after the other behind one call (the rounds loop of apps/bfs/bfs.cpp:36-50): d_levels is reused
 * and holds the levels of the last source afterwards; stats (optional) has `count` entries.  Stops at the first failing traversal. */
int vgl_hip_bfs_run_batch(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *sources, int32_t count, int mode, int32_t *d_levels,
                          vgl_hip_bfs_stats *stats);
/* Optional graph preparation for repeated top-down traversals (the counterpart of the reference's offline graph import,
 * vgl_graph.hpp:57-68): lays the outgoing edges out for the blocked advance (4 bytes per edge kept, a radix sort of the edges once).
 * Afterwards vgl_hip_bfs_run expands the levels that hold at least a tenth of the edges (VGL_BFS_BLOCKED_SHARE) as a blocked pass --
 * one bit per edge travels from the source's block to the destination's, 4.2 bytes per edge streamed -- instead of
 * bfs.hpp:28-36's per-edge probe of levels[dst].  Levels are identical.  The handle must own all rows. */
int vgl_hip_bfs_prepare_blocked(vgl_hip_ctx *ctx, vgl_hip_graph *g);

typedef struct {
    int32_t iterations;
    int64_t edges_relaxed;     /* edges streamed by relax kernels */
    int64_t algorithmic_bytes; /* 12*edges_relaxed + 28*V*iterations */
    int32_t push_steps, pull_steps;   /* super-steps by direction (iterations = their sum) */
} vgl_hip_sssp_stats;
#define VGL_HIP_SSSP_ALL_ACTIVE 0        /* vgl_dijkstra_all_active_push, shortest_paths.hpp:85-163: every iteration streams all edges */
#define VGL_HIP_SSSP_ACTIVE_TILES 1      /* same fixed point, skips edge tiles whose sources did not change */
/* 2 is the bucketed schedule (vgl_hip_sssp_run_delta below; the Python harness uses the number for it) */
#define VGL_HIP_SSSP_PULL 3              /* vgl_dijkstra_all_active_pull, shortest_paths.hpp:169-292: every vertex takes the minimum over its
                                            incoming edges; here a blocked gather / LDS-minimum pass, no scattered stores, no per-edge L2 line */
#define VGL_HIP_SSSP_DIRECTION_OPT 4     /* push over the compacted frontier of the rows that changed (atomic minima) while they own few edges,
                                            pull while they own many; the switch is on the share of edges whose source changed in the last
                                            super-step (VGL_SSSP_PULL_SHARE, 0.2) */
int vgl_hip_sssp_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, int32_t source, int mode,
                     float *d_dist, vgl_hip_sssp_stats *stats);
/* SSWP::vgl_dijkstra, algorithms/sswp/widest_paths.hpp:5-76 (single-source widest paths): widths[source] = FLT_MAX, others 0;
 * width[dst] = max(width[dst], min(width[src], capacity)) to the fixed point.  Same kernel and modes as vgl_hip_sssp_run with the
 * (max, min) path algebra; only min / max of the inputs occur, so the result is bit-identical to the reference (and to its
 * sequential checker, seq_widest_paths.hpp:5-64).  d_capacities is indexed like the outgoing CSR (global_edge_pos). */
/* super-step pieces of the widest-path algorithm for shards (exchange between steps: allreduce(MAX) of the widths) */
int vgl_hip_sswp_init(vgl_hip_ctx *ctx, int32_t V, int32_t source, float *d_widths);
int vgl_hip_sswp_relax_owned(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_capacities, float *d_widths, int *changed);
int vgl_hip_sswp_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_capacities, int32_t source, int mode,
                     float *d_widths, vgl_hip_sssp_stats *stats);
/* The pull steps work on a blocked copy of (outgoing adjacency, edge values) -- vgl_blocked.h -- built once per (graph, weights)
 * like the reference's graph import and reusable for any number of sources; vgl_hip_sssp_run / vgl_hip_sswp_run with mode PULL or
 * DIRECTION_OPT = create + run + destroy.  mode: VGL_HIP_SSSP_PULL or VGL_HIP_SSSP_DIRECTION_OPT.
 * A plan holds a reordered COPY of the edge values: the caller must not rewrite d_weights in place while the plan exists (rebuild the plan
 * after set_all_random or any other writer), and must destroy the plan before the graph handle; a plan is refused for any other handle. */
typedef struct vgl_hip_sssp_pull_plan vgl_hip_sssp_pull_plan;
int vgl_hip_sssp_pull_plan_create(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan **out);
int vgl_hip_sssp_pull_plan_destroy(vgl_hip_ctx *ctx, vgl_hip_sssp_pull_plan *plan);
/* edges laid out, how many of them as fused tiles (dense pairs of 16384-id blocks: both windows in LDS, 8 bytes per edge streamed instead of
 * 16), bytes one pull pass streams (pad entries included) and bytes the plan keeps resident */
int vgl_hip_sssp_pull_plan_info(vgl_hip_sssp_pull_plan *plan, int64_t *edges, int64_t *fused_edges, int64_t *streamed_bytes_per_pass, int64_t *plan_bytes);
/* one all-edges relaxation through the plan (values of the pass start; the relax of shortest_paths.hpp:123-133 over every edge);
 * *changed = 1 when a distance decreased.  Synchronises. */
int vgl_hip_sssp_pull_pass(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_sssp_pull_plan *plan, float *d_dist, int *changed);
int vgl_hip_sssp_run_pull(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan *plan, int32_t source,
                          int mode, float *d_dist, vgl_hip_sssp_stats *stats);
int vgl_hip_sswp_run_pull(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_capacities, vgl_hip_sssp_pull_plan *plan, int32_t source,
                          int mode, float *d_widths, vgl_hip_sssp_stats *stats);

/* Same operators and bit-identical distances, bucketed schedule (delta-stepping with a light/heavy edge split): light
 * edges (w < delta) of a vertex are relaxed whenever it improves inside the current distance bucket, heavy edges once the
 * bucket has settled.  Cuts the per-edge dist[dst] gathers from ~5 E (Bellman-Ford) to ~1.2 E.  stats->iterations = relax
 * launches.  delta > 0, in the unit of the weights (weights in [0,100): 10..25 works well). */
int vgl_hip_sssp_run_delta(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, int32_t source, float delta,
                           float *d_dist, vgl_hip_sssp_stats *stats);
/* The bucketed schedule works on a plan: a copy of the adjacency + weights in which every row's edges are stably
 * partitioned light-first (w < delta), built once per (graph, weights, delta) -- preprocessing in the sense of the
 * reference's graph import, reusable for any number of sources.  run_delta above = create + run + destroy. */
typedef struct vgl_hip_sssp_plan vgl_hip_sssp_plan;
int vgl_hip_sssp_plan_create(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, float delta, vgl_hip_sssp_plan **out);
int vgl_hip_sssp_plan_destroy(vgl_hip_ctx *ctx, vgl_hip_sssp_plan *plan);
int vgl_hip_sssp_run_plan(vgl_hip_ctx *ctx, vgl_hip_graph *g, vgl_hip_sssp_plan *plan, int32_t source, float *d_dist,
                          vgl_hip_sssp_stats *stats);

typedef struct {
    int32_t iterations;
    double ranks_sum;          /* reduce_ranks_sum of the last iteration (pr.hpp:130-134) */
    int64_t algorithmic_bytes; /* (8*E + 28*V) * iterations */
} vgl_hip_pr_stats;
/* vgl_page_rank, pr.hpp:7-149 (f32, d = 0.85, fixed iteration count, reversed-graph definition).
 * d_indeg_noloops: int32[V] = in-degree minus self loops (pr.hpp:31-65), or NULL to have it computed. */
int vgl_hip_pr_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *d_indeg_noloops, int iterations,
                   float *d_ranks, vgl_hip_pr_stats *stats);
/* Two evaluations of the per-vertex sum (north star: within 1e-6 relative of the reference):
 *   EXACT_ORDER  f32 `+=` chain in adjacency order, bit-identical to seq_page_rank (seq_pr.hpp:81-96); one random L2 line per edge
 *   BLOCKED      contributions gathered from and summed in 128 KiB LDS windows (vgl_blocked.h): 12 B/edge of streamed HBM traffic, no
 *                random line per edge; the sums are exact (64-bit fixed point, rounded to f32 once) and therefore independent of any
 *                order -- they differ from the chain by the CHAIN's rounding error, ~sqrt(n) * 3e-8 for a row of n entries
 *   AUTO         BLOCKED when the graph stores >= 2^25 edges and no row is longer than 256 entries (uniform-random inputs), else
 *                EXACT_ORDER; what vgl_hip_pr_run uses (VGL_PR_MODE=0|1 overrides) */
#define VGL_HIP_PR_EXACT_ORDER 0
#define VGL_HIP_PR_BLOCKED 1
#define VGL_HIP_PR_AUTO 2
int vgl_hip_pr_run_mode(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *d_indeg_noloops, int iterations, int mode,
                        float *d_ranks, vgl_hip_pr_stats *stats);
/* Graph preparation for PageRank / Shiloach-Vishkin (the counterpart of the reference's offline import): builds NOW what the first
 * vgl_hip_pr_run / vgl_hip_cc_run would otherwise build inside its first call -- the blocked layout (a radix sort of the edges, tens of ms
 * for 10^9 edges plus the allocations) or the hub schedule of the ordered pull.  *resolved_mode (optional): what AUTO resolved to.  The
 * default entry point vgl_hip_pr_run uses AUTO: bit-identical to seq_page_rank below 2^25 stored edges or when a row is longer than 256
 * entries, exact per-vertex sums (<= 1e-6 of the chain on such graphs, NOT bit-identical) otherwise; pass VGL_HIP_PR_EXACT_ORDER to
 * vgl_hip_pr_run_mode for the reference's evaluation order at any size. */
int vgl_hip_pr_prepare(vgl_hip_ctx *ctx, vgl_hip_graph *g, int mode, int *resolved_mode);
int vgl_hip_cc_prepare(vgl_hip_ctx *ctx, vgl_hip_graph *g);
/* Graph preparation for the path algorithms (Bellman-Ford / widest paths; counterpart of the reference's import, which derives every edge-array layout
 * from one permutation, csr_edges_array.hpp:31-40): builds the blocked STRUCTURE of the outgoing CSR once per graph -- one radix sort of the edges by
 * block pair, the CSR position behind every value slot kept.  Afterwards a pull plan for ANY weights array (vgl_hip_sssp_pull_plan_create) is one
 * gather pass, and vgl_hip_sssp_run(ALL_ACTIVE) -- the reference's schedule, every edge in every super-step -- runs as blocked passes (same fixed
 * point, same f32 bits).  Needs a handle that owns all rows. */
int vgl_hip_sssp_prepare(vgl_hip_ctx *ctx, vgl_hip_graph *g);

typedef struct {
    int32_t hook_passes;
    int64_t algorithmic_bytes; /* (8*E + 12*V) per hook pass + 12*V per jump pass */
} vgl_hip_cc_stats;
/* SCC::vgl_forward_backward, algorithms/scc/scc.hpp (checker SCC::seq_tarjan, seq_scc.hpp): strongly connected components of a
 * directed graph.  d_comp[v] = smallest vertex id of v's component: the canonical form of the partition (the reference's labels are
 * arbitrary counters and its test compares partitions, verify_results.h equal_components).  Needs the incoming CSR. */
typedef struct {
    int32_t trim_rounds;             /* vertex passes that removed trivial components */
    int32_t forward_backward_steps;  /* pivot reach steps (0 or 1: the big component) */
    int32_t colour_rounds;           /* colour-class rounds for the remaining components */
    int32_t edge_passes;             /* all-edges passes of those rounds (inactive tiles skipped) */
} vgl_hip_scc_stats;
int vgl_hip_scc_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_scc_stats *stats);

/* HITS::vgl_hits, algorithms/hits/hits.hpp:5-100 (f64, apps/hits/hits.cpp:13): auth = hub = 1, then `steps` times
 *   auth[v] = sum of hub over the in-neighbours, auth /= ||auth||_2, hub[v] = sum of auth over the out-neighbours, hub /= ||hub||_2.
 * Per-vertex sums run in adjacency order (the sequential checker's order, hits.hpp:117-160); norms are folded in a fixed order.
 * Needs the incoming CSR.  Deterministic; agrees with seq_hits to ~1e-15 relative. */
int vgl_hip_hits_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, int steps, double *d_auth, double *d_hub);

/* vgl_shiloach_vishkin, shiloach_vishkin.hpp:7-88: labels = min id that reaches each vertex */
int vgl_hip_cc_run(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats);
/* Same labels for SYMMETRIC graphs only (every edge stored in both directions, as the reference's cc app builds its input,
 * apps/cc/cc.cpp:24-36 UNDIRECTED_GRAPH): the fixed point of the hook/jump loop is then "smallest id of the connected component",
 * which a min-id union-find reaches without sweeping all edges repeatedly -- two sampled neighbours per vertex, then only the
 * rows outside the largest tree look at their edges.  The caller vouches for the symmetry; on a directed graph the labels
 * are those of the weakly-connected components of the stored edges, NOT vgl_hip_cc_run's.  stats->hook_passes counts the
 * link passes (sampling rounds + 1), algorithmic_bytes what those passes had to touch. */
int vgl_hip_cc_run_symmetric(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats);

/* ---- super-step pieces for the edge-cut multi-GPU path (one process per GPU; the exchange between steps is an
 *      RCCL collective issued by the host side, replacing common/mpi_exchange.hpp:110-150,222-271) ---- */
int vgl_hip_bfs_init(vgl_hip_ctx *ctx, int32_t V, int32_t source, int32_t *d_levels);
/* expand the owned part of level `level`: frontier = owned rows with levels == level.  Writes levels[dst] = level+1
 * anywhere in the replicated array.  d_visited_bits: replicated visited bitmap (V bits) or NULL to have it rebuilt from
 * levels.  local_frontier/local_edges are host outputs (the frontier is compacted before the expand, which is left enqueued). */
int vgl_hip_bfs_step_top_down(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_levels, int32_t level,
                              const uint64_t *d_visited_bits, int64_t *local_frontier, int64_t *local_edges);
/* bottom-up step over the owned rows (needs the incoming CSR): unvisited owned vertices with an in-neighbour in the frontier
 * bitmap get levels = level+1.  d_next_bits (V bits) receives exactly this rank's discoveries (other words zero), ready for
 * the bitmap exchange.  found / probed (optional, synchronise): discovered vertices, adjacency entries examined. */
/* top-down step driven by the replicated bitmaps: the owned part of the frontier is read from d_front_bits (V/64 words instead
 * of a scan of levels), every vertex this shard discovers gets levels[v] = level+1 and its bit in d_next_bits (cleared here,
 * full length: destinations live in any shard).  d_next_bits is what the shard contributes to the exchange. */
int vgl_hip_bfs_step_top_down_bits(vgl_hip_ctx *ctx, vgl_hip_graph *graph, int32_t *d_levels, int32_t level,
                                   const uint64_t *d_visited_bits, const uint64_t *d_front_bits, uint64_t *d_next_bits,
                                   int64_t *local_frontier, int64_t *local_edges);
/* out[w] = OR over p < parts of in[p * words + w]  (merging the bitmap slices a rank received in the two-phase exchange) */
int vgl_hip_bitmap_or_parts(vgl_hip_ctx *ctx, int64_t words, int parts, const uint64_t *d_in, uint64_t *d_out);
int vgl_hip_bfs_step_bottom_up(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_levels, int32_t level,
                               const uint64_t *d_visited_bits, const uint64_t *d_front_bits, uint64_t *d_next_bits,
                               int64_t *found, int64_t *probed);
/* bitmap (V bits, little-endian within uint64 words) of vertices with d_levels == level */
int vgl_hip_levels_to_bitmap(vgl_hip_ctx *ctx, int32_t V, const int32_t *d_levels, int32_t level, uint64_t *d_bits);
/* OR `parts` bitmaps (each V/64 words, contiguous) and set levels[v] = level where a bit is set and v is unvisited.
 * Optional replicated state for the next direction decision: d_visited_bits |= new frontier, d_front_bits = new frontier,
 * d_degrees (int32[V] out-degrees of ALL vertices) -> *newly_degree = sum of the new frontier's out-degrees.
 * *newly = number of vertices that now have d_levels == level.  Synchronises. */
int vgl_hip_bfs_apply_bitmaps(vgl_hip_ctx *ctx, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                              uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly,
                              int64_t *newly_degree);
/* vgl_hip_bfs_apply_bitmaps for a rank that keeps levels only for the rows it owns: the replicated bitmaps are updated for every vertex
 * (any rank may probe any vertex), d_levels / the two results only for [own_begin, own_end) (multiples of 64) -- the per-vertex part
 * of the merge no longer grows with the whole graph on every rank; the caller adds the two results over the ranks. */
int vgl_hip_bfs_apply_bitmaps_owned(vgl_hip_ctx *ctx, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                                    uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int32_t own_begin,
                                    int32_t own_end, int64_t *newly_owned, int64_t *newly_owned_degree);
/* Sparse exchange of tiny levels (the id-list counterpart of the bitmap exchange of common/mpi_exchange.hpp:222-271):
 * d_out[0] = number of set bits of the bitmap (may exceed cap), d_out[1 .. 1+cap) = ids of (the first cap of) them, unordered.
 * Asynchronous on the context's stream. */
int vgl_hip_bitmap_to_ids(vgl_hip_ctx *ctx, int64_t words, const uint64_t *d_bits, int32_t cap, int32_t *d_out);
/* vgl_hip_bfs_apply_bitmaps for `parts` id lists of that layout (stride 1 + cap, every count <= cap): unvisited listed vertices get
 * d_levels = level, their visited bits are set and d_front_bits becomes exactly the set of them.  Synchronises. */
int vgl_hip_bfs_apply_ids(vgl_hip_ctx *ctx, int32_t V, int parts, int32_t cap, const int32_t *d_lists, int32_t *d_levels, int32_t level,
                          uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly,
                          int64_t *newly_degree);
int vgl_hip_sssp_init(vgl_hip_ctx *ctx, int32_t V, int32_t source, float *d_dist);
/* one all-active push relaxation over the owned rows; *changed = 1 if any distance decreased. Synchronises. */
int vgl_hip_sssp_relax_owned(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_weights, float *d_dist, int *changed);
int vgl_hip_cc_init(vgl_hip_ctx *ctx, int32_t V, int32_t *d_comp);
int vgl_hip_cc_hook_owned(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_comp, int *changed);
int vgl_hip_cc_jump(vgl_hip_ctx *ctx, int32_t V, int32_t *d_comp);
/* PageRank pieces: prepare (contrib = old*rdeg, dangling sum over ALL vertices) and pull over the owned rows only;
 * ranks of owned rows are written, the caller all-gathers owned slices (EXCHANGE_PRIVATE_DATA, pr.hpp:127). */
int vgl_hip_pr_setup(vgl_hip_ctx *ctx, int32_t V, const int32_t *d_indeg_noloops, float *d_ranks, float *d_rdeg);
/* The sum-type all-active advance of PR::vgl_page_rank (algorithms/pr/pr.hpp:105-124: `page_ranks[src] += contribution[dst]` over every edge
 * src -> dst with dst != src) with the caller's own arrays: d_sums[src] = sum over those edges of d_values[dst], for the rows the handle owns.
 * Runs as the blocked pass (values read from LDS windows, exact 64-bit fixed-point accumulation, rounded to f32 once: the same bits for any
 * schedule; it differs from an f32 `+=` chain in adjacency order by the chain's own rounding).  d_values must be non-negative and every
 * per-vertex sum at most sum_bound (the unit of the fixed point is derived from it; PageRank contributions: 1).  The layout is the one
 * vgl_hip_pr_prepare(BLOCKED) builds (built here on first use).  Asynchronous on the context's stream. */
int vgl_hip_sum_over_edges_f32(vgl_hip_ctx *ctx, vgl_hip_graph *g, const float *d_values, float sum_bound, float *d_sums);
int vgl_hip_pr_iteration_owned(vgl_hip_ctx *ctx, vgl_hip_graph *g, const int32_t *d_indeg_noloops, const float *d_rdeg,
                               float *d_ranks, float *d_contrib_scratch);
/* "Recently changed" exchange of a replicated 4-byte vertex array (EXCHANGE_RECENTLY_CHANGED, common/mpi_exchange.hpp:110-150): instead
 * of all-reducing V entries per super-step every rank sends the (index, value) pairs of the entries its step changed.
 * diff_to_pairs: d_out[0] = number of i < n with d_before[i] != d_after[i] (may exceed cap), then (index, value bits) pairs of the first
 * min(count, cap) of them, unordered; d_out holds 1 + 2 * cap int32.  Asynchronous.
 * apply_pairs: `parts` such lists, `stride` int32 apart; every pair is merged into d_values with min (take_min != 0: SSSP distances, CC
 * labels) or max (SSWP widths) on the 4-byte patterns (all values are non-negative); list `skip_part` (this rank's own, or -1) is
 * skipped.  changed (optional, synchronises): 1 if an entry of d_values moved. */
int vgl_hip_diff_to_pairs_u32(vgl_hip_ctx *ctx, int32_t n, const void *d_before, const void *d_after, int32_t cap, int32_t *d_out);
int vgl_hip_apply_pairs_u32(vgl_hip_ctx *ctx, int parts, int64_t stride, int skip_part, const int32_t *d_lists, int take_min, int32_t n,
                            void *d_values, int *changed);
/* in-degree without self loops from an out-CSR shard (adds into d_indeg; zero it first; allreduce(sum) across shards) */
int vgl_hip_indegree_noloops_add(vgl_hip_ctx *ctx, vgl_hip_graph *g, int32_t *d_indeg);

/* ---- multi-GPU behind the boundary: communicator, exchanges and the super-step loops (one process per GPU).
 *      Replaces GraphAbstractions::exchange_vertices_array (common/graph_abstractions.h:157-168) and its MPI implementation
 *      common/mpi_exchange.hpp:110-150 (EXCHANGE_RECENTLY_CHANGED), :156-187 (EXCHANGE_ALL with a merge operator), :222-271
 *      (EXCHANGE_PRIVATE_DATA); the library_data MPI rank / size of vgl_runtime/helpers/library_data/library_data.h.
 *      Transport RCCL: collectives over xGMI, enqueued on the context's stream -- kernels and collectives of a super-step are
 *      stream-ordered, the host waits only where it has to decide something (one pinned-memory poll per super-step).
 *      Transport HOSTED: the same collectives staged through a POSIX shared-memory segment; ranks are processes of one host and
 *      may share one GPU (RCCL refuses two ranks on one device) -- rehearsals and multi-rank tests on fewer GPUs than ranks. ---- */
typedef struct vgl_hip_comm vgl_hip_comm;
#define VGL_HIP_COMM_ID_BYTES 128
#define VGL_HIP_COMM_RCCL 0
#define VGL_HIP_COMM_HOSTED 1
#define VGL_HIP_COMM_PEER 2
/* rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means (file, socket, MPI, torch store) */
int vgl_hip_comm_unique_id(void *id_out);
int vgl_hip_comm_create(vgl_hip_ctx *ctx, int rank, int world, const void *unique_id, vgl_hip_comm **out);
/* name: shared-memory object name common to the ranks ("/vgl_job42"); slot_bytes: staging capacity per rank (larger payloads go in pieces) */
int vgl_hip_comm_create_hosted(vgl_hip_ctx *ctx, int rank, int world, const char *name, size_t slot_bytes, vgl_hip_comm **out);
/* Transport PEER (round 4; what the reference hand-rolls with MPI point-to-point messages, vgl_compute_api/common/mpi_exchange.hpp:110-187,222-271):
 * every rank owns a window in device memory that the other ranks map (hipIpc; ranks = GPUs of one node over xGMI, or processes / threads
 * sharing one GPU) and write into from kernels on their own streams; arrival and consumption flags live in the windows; the exchanges of
 * a group share one flag round.  No collective library and no host in the data path.  name: shared-memory object used for the set-up
 * handshake only; window_bytes: capacity of one of the two halves of a window (larger payloads go in pieces).  Fails -- on every rank alike --
 * when a window cannot be mapped by a peer: fall back to vgl_hip_comm_create (RCCL). */
int vgl_hip_comm_create_peer(vgl_hip_ctx *ctx, int rank, int world, const char *name, size_t window_bytes, vgl_hip_comm **out);
/* bound of every in-kernel flag wait of the PEER transport from now on (ms; <= 0: the default, 20 s).  The variable VGL_PEER_TIMEOUT_MS is read once,
 * when the communicator is created; a caller that probes the transport with a short bound sets the working bound here afterwards.  Other transports: no-op. */
int vgl_hip_comm_set_timeout_ms(vgl_hip_comm *comm, double ms);
/* tells the other ranks of a hosted / peer communicator that this rank gives up: their next barrier fails at once instead of after its timeout */
int vgl_hip_comm_abort(vgl_hip_comm *comm);
int vgl_hip_comm_destroy(vgl_hip_comm *comm);
int vgl_hip_comm_info(vgl_hip_comm *comm, int *rank, int *world, int *transport);
int vgl_hip_comm_barrier(vgl_hip_comm *comm);            /* drains the stream, meets the other ranks */
/* EXCHANGE_ALL with the merge operators the algorithms use (shortest_paths.hpp:136-141 min_op; widest paths: max; pr.hpp:58 sum):
 * in-place all-reduce of a replicated device array, asynchronous on the context's stream */
int vgl_hip_exchange_allreduce_min_i32(vgl_hip_comm *comm, int32_t *d_values, int64_t n);
int vgl_hip_exchange_allreduce_min_f32(vgl_hip_comm *comm, float *d_values, int64_t n);
int vgl_hip_exchange_allreduce_max_f32(vgl_hip_comm *comm, float *d_values, int64_t n);
int vgl_hip_exchange_allreduce_sum_i32(vgl_hip_comm *comm, int32_t *d_values, int64_t n);
int vgl_hip_exchange_allreduce_sum_i64(vgl_hip_comm *comm, int64_t *d_values, int64_t n);
int vgl_hip_exchange_allreduce_sum_f32(vgl_hip_comm *comm, float *d_values, int64_t n);
int vgl_hip_exchange_allreduce_sum_f64(vgl_hip_comm *comm, double *d_values, int64_t n);
/* d_recv[p * bytes .. ) = rank p's d_send; asynchronous */
int vgl_hip_exchange_allgather(vgl_hip_comm *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank);
/* EXCHANGE_PRIVATE_DATA (mpi_exchange.hpp:222-271, MPI_Allgatherv in place): rank p owns elements [bounds[p], bounds[p+1]) of the
 * replicated array (host array of world+1 entries, the same on every rank); afterwards every rank holds every owner's slice */
int vgl_hip_exchange_allgather_slices(vgl_hip_comm *comm, void *d_array, const int64_t *bounds_host, int elem_bytes);
/* every rank ends with the OR of all ranks' bitmaps (in place): all-to-all of the word slices, OR, all-gather -- 2 x words x 8 bytes per
 * rank instead of world x words x 8 */
int vgl_hip_exchange_bitmap_or(vgl_hip_comm *comm, uint64_t *d_bits, int64_t words);
/* EXCHANGE_RECENTLY_CHANGED in one call (diff -> counts -> lists -> merge; mpi_exchange.hpp:110-150): d_values is this rank's copy after its
 * super-step, d_before the copy before it; afterwards d_values holds the merge (min or max on the 4-byte patterns of non-negative values)
 * of every rank's changes.  One small all-gather when every rank changed at most 2048 entries (the counts ride in the payload), a second,
 * sized one otherwise, the whole-array all-reduce when some rank changed more than n / (2 world).  *changed_anywhere (synchronises): 1 if
 * any rank changed anything -- the loop condition of shortest_paths.hpp:143-152 without a flag reduction. */
int vgl_hip_exchange_changed_u32(vgl_hip_comm *comm, int32_t n, const void *d_before, void *d_values, int take_min, int *changed_anywhere);

/* The super-step loops over edge-cut shards (g owns rows [row_begin, row_end) of a graph of V vertices; vertex arrays are replicated,
 * V entries on every rank).  comm == NULL or a world of one: no exchange, same code path.  Results are bit-identical to the single-GPU
 * drivers (fixed points / owner-computed sums).
 *   bfs : direction-optimising (mode DIRECTION_OPT needs the incoming CSR of the owned rows and global_edges = E of the whole graph) or
 *         top-down; every rank must own a 64-aligned row range.  d_levels is complete on the OWNED rows when it returns; pass
 *         gather_levels != 0 to have the slices all-gathered (EXCHANGE_PRIVATE_DATA) so that every rank holds all levels.
 *   sssp / sswp : all-active push over the owned rows + changed-entries exchange (min / max)
 *   cc  : Shiloach-Vishkin hook over the owned rows + changed-entries exchange (min) + replicated pointer jumping
 *   pr  : owner-computes pull + all-gather of the owned slices; mode as vgl_hip_pr_run_mode, AUTO is resolved from the GLOBAL edge count
 *         and the GLOBAL longest row so that every rank takes the same path.  The in-degrees minus self loops of all vertices (pr.hpp:31-65)
 *         are counted and summed over the ranks on the first call with a graph handle and kept with it (the graph does not change) */
int vgl_hip_bfs_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, int32_t source, int mode, int64_t global_edges,
                            int gather_levels, int32_t *d_levels, vgl_hip_bfs_stats *stats);
int vgl_hip_sssp_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, const float *d_weights, int32_t source,
                             float *d_dist, vgl_hip_sssp_stats *stats);
int vgl_hip_sswp_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, const float *d_capacities, int32_t source,
                             float *d_widths, vgl_hip_sssp_stats *stats);
int vgl_hip_cc_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats);
int vgl_hip_pr_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, int iterations, int mode, float *d_ranks,
                           vgl_hip_pr_stats *stats);
/* HITS over the shards (algorithms/hits/hits.hpp:32-91 with the exchanges at :52 and :79): replicated f64 authority / hub arrays, the owned
 * rows pulled per rank, one f64 all-reduce (sum of squares) and one all-gather of the owned slices per half step */
int vgl_hip_hits_run_sharded(vgl_hip_ctx *ctx, vgl_hip_comm *comm, vgl_hip_graph *g, int steps, double *d_auth, double *d_hub);
/* exchange statistics of the last *_run_sharded on this communicator: collectives issued, bytes this rank received, super-steps that
 * used pair lists / the whole-array all-reduce / id lists (BFS) */
typedef struct {
    int64_t collectives, bytes_received;
    int32_t list_steps, dense_steps, sparse_levels, exchanges;      /* exchanges: flag rounds of the PEER transport (the collectives of a group share one) */
} vgl_hip_exchange_stats;
int vgl_hip_comm_stats(vgl_hip_comm *comm, vgl_hip_exchange_stats *out);

/* ---- kernel timing hooks for bench.py's roofline line: when enabled every launch of the named dominant kernels
 *      is bracketed by hipEvents on the context stream; totals are read back afterwards. ---- */
int vgl_hip_timing_enable(vgl_hip_ctx *ctx, int enable);
/* restrict the bracketing to ONE kernel name (NULL or "" = all): two event records cost ~4-5 us of stream time per launch, which is
 * 10 % of a BFS traversal when every kernel is bracketed -- the timed region of bench.py brackets only the kernel it reports */
int vgl_hip_timing_only(vgl_hip_ctx *ctx, const char *kernel_name);
/* of the launches that would be bracketed only every stride-th is (1 = all): sampling instead of perturbing a launch-bound loop */
int vgl_hip_timing_stride(vgl_hip_ctx *ctx, int stride);
int vgl_hip_timing_reset(vgl_hip_ctx *ctx);
/* kernel_name: "bfs_bottom_up", "bfs_top_down", "gnf", "sssp_relax", "pr_pull", "cc_hook"; returns launches and total ms */
int vgl_hip_timing_get(vgl_hip_ctx *ctx, const char *kernel_name, int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif
