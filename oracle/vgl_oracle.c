/*
 * vgl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See vgl_oracle.h.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared  (no -ffast-math: the PageRank and
 * SSSP parity claims depend on IEEE f32 evaluation in source order).
 */
#include "vgl_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* RNG + generators                                                                           */
/* ------------------------------------------------------------------------------------------ */

uint64_t vgo_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

uint32_t vgo_relabel(uint32_t v, int scale, uint64_t seed)
{
    if (scale <= 0) return v;
    const uint32_t mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    const int sh = (scale + 1) / 2;
    uint32_t x = v & mask;
    for (int r = 0; r < 3; r++) {
        uint32_t m = (uint32_t)vgo_splitmix64(seed + 0x100 + (uint64_t)r) | 1u; /* odd => bijective mod 2^s */
        uint32_t k = (uint32_t)(vgo_splitmix64(seed + 0x200 + (uint64_t)r) >> 32);
        x = (x * m) & mask;
        x ^= x >> sh;
        x = (x + k) & mask;
    }
    return x;
}

/* graph_generation.hpp:94-187: start in the centre of the V x V matrix, (scale-1) quadrant draws
 * with probabilities a/b/c/d percent moving (row,col) by +-2^(n-(i+1)), then two coin flips
 * each subtracting 1.  `from` = row, `to` = col.  Duplicates and self-loops are kept. */
void vgo_gen_rmat(int scale, int64_t first_edge, int64_t count, uint64_t seed,
                  int a, int b, int c, int d, int relabel, int32_t *src, int32_t *dst)
{
    (void)d;
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < count; t++) {
        const uint64_t idx = (uint64_t)(first_edge + t);
        const uint64_t key = vgo_splitmix64(seed ^ vgo_splitmix64(idx));
        uint32_t x = 1u << (scale - 1), y = 1u << (scale - 1);
        uint64_t word = 0;
        for (int i = 1; i < scale; i++) {
            const int q = i - 1;
            if ((q & 1) == 0) word = vgo_splitmix64(key + (uint64_t)(q >> 1));
            const uint32_t r32 = (q & 1) ? (uint32_t)(word >> 32) : (uint32_t)word;
            const uint32_t p = r32 % 100u;
            const uint32_t step = 1u << (scale - (i + 1));
            if (p < (uint32_t)a)                { x -= step; y -= step; }
            else if (p < (uint32_t)(a + b))     { x -= step; y += step; }
            else if (p < (uint32_t)(a + b + c)) { x += step; y -= step; }
            else                                { x += step; y += step; }
        }
        const uint64_t flips = vgo_splitmix64(key + 64);
        if ((flips & 1) == 0) x--;
        if ((flips & 2) == 0) y--;
        if (relabel) { x = vgo_relabel(x, scale, seed); y = vgo_relabel(y, scale, seed); }
        src[t] = (int32_t)x;
        dst[t] = (int32_t)y;
    }
}

/* graph_generation.hpp:5-51: src, dst i.i.d. uniform in [0, V) */
void vgo_gen_uniform(int scale, int64_t first_edge, int64_t count, uint64_t seed,
                     int32_t *src, int32_t *dst)
{
    const uint32_t mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < count; t++) {
        const uint64_t idx = (uint64_t)(first_edge + t);
        const uint64_t h = vgo_splitmix64((seed + 0x5151ULL) ^ vgo_splitmix64(idx));
        src[t] = (int32_t)((uint32_t)(h >> 32) & mask);
        dst[t] = (int32_t)((uint32_t)h & mask);
    }
}

void vgo_gen_weights(int64_t first_edge, int64_t count, uint64_t seed, float *w)
{
    #pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < count; t++) {
        const uint64_t idx = (uint64_t)(first_edge + t);
        const uint64_t h = vgo_splitmix64((seed + 0x7777ULL) ^ vgo_splitmix64(idx));
        w[t] = (float)(uint32_t)(h >> 40) * (100.0f / 16777216.0f); /* 24 random bits -> [0,100) */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* graph build                                                                                */
/* ------------------------------------------------------------------------------------------ */

void vgo_coo_to_csr(int32_t V, int64_t E, const int32_t *src, const int32_t *dst,
                    int64_t *rowptr, int32_t *adj, int64_t *perm)
{
    memset(rowptr, 0, sizeof(int64_t) * ((size_t)V + 1));
    for (int64_t e = 0; e < E; e++) rowptr[src[e] + 1]++;
    for (int32_t v = 0; v < V; v++) rowptr[v + 1] += rowptr[v];
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));
    memcpy(cursor, rowptr, sizeof(int64_t) * (size_t)V);
    for (int64_t e = 0; e < E; e++) {            /* input order => stable */
        const int64_t p = cursor[src[e]]++;
        adj[p] = dst[e];
        if (perm) perm[p] = e;
    }
    free(cursor);
}

typedef struct { int64_t deg; int32_t id; } vgo_degid;
static int vgo_cmp_degid(const void *pa, const void *pb)
{
    const vgo_degid *a = (const vgo_degid *)pa, *b = (const vgo_degid *)pb;
    if (a->deg != b->deg) return (a->deg > b->deg) ? -1 : 1; /* degree descending */
    return (a->id > b->id) - (a->id < b->id);                 /* id ascending == stable */
}
void vgo_degree_renumber(int32_t V, const int64_t *rowptr, int32_t *fwd, int32_t *bwd)
{
    vgo_degid *t = (vgo_degid *)malloc(sizeof(vgo_degid) * (size_t)(V > 0 ? V : 1));
    for (int32_t v = 0; v < V; v++) { t[v].deg = rowptr[v + 1] - rowptr[v]; t[v].id = v; }
    qsort(t, (size_t)V, sizeof(vgo_degid), vgo_cmp_degid);
    for (int32_t s = 0; s < V; s++) { bwd[s] = t[s].id; fwd[t[s].id] = s; }
    free(t);
}

/* ------------------------------------------------------------------------------------------ */
/* BFS                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* generate_new_frontier_worker(CSRGraph&) (multicore/generate_new_frontier.hpp:113-164):
 * flags from predicate, then ascending-id compaction (copy_if.hpp:128-191). */
static int64_t vgo_gnf_level(int32_t V, const int32_t *levels, int32_t want, int32_t *ids, int parallel)
{
#ifdef _OPENMP
    if (parallel) {
        const int nt = omp_get_max_threads();
        int64_t *cnt = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
        #pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const int64_t lo = (int64_t)V * t / nt, hi = (int64_t)V * (t + 1) / nt;
            int64_t c = 0;
            for (int64_t v = lo; v < hi; v++) c += (levels[v] == want);
            cnt[t + 1] = c;
            #pragma omp barrier
            #pragma omp single
            { for (int i = 0; i < nt; i++) cnt[i + 1] += cnt[i]; }
            int64_t p = cnt[t];
            for (int64_t v = lo; v < hi; v++) if (levels[v] == want) ids[p++] = (int32_t)v;
        }
        const int64_t total = cnt[nt];
        free(cnt);
        return total;
    }
#endif
    (void)parallel;
    int64_t n = 0;
    for (int32_t v = 0; v < V; v++) if (levels[v] == want) ids[n++] = v;
    return n;
}

void vgo_bfs_top_down(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t source,
                      int32_t *levels, vgo_bfs_stats *st, int parallel)
{
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    /* bfs.hpp:12-21 init_levels: source = FIRST_LEVEL_VERTEX (1), others UNVISITED_VERTEX (-1) */
    #pragma omp parallel for schedule(static) if (parallel)
    for (int32_t v = 0; v < V; v++) levels[v] = -1;
    levels[source] = 1;
    ids[0] = source;
    int64_t fsize = 1;
    int32_t cur = 1;
    vgo_bfs_stats s = {0, 0, 0, 0};
    while (fsize > 0) {                                   /* bfs.hpp:26 */
        int64_t m = 0;
        /* scatter over the sparse frontier (advance_worker.hpp:112-139), edge_op bfs.hpp:28-36 */
        #pragma omp parallel for schedule(guided, 1024) reduction(+ : m) if (parallel)
        for (int64_t i = 0; i < fsize; i++) {
            const int32_t u = ids[i];
            const int64_t b = rowptr[u], e = rowptr[u + 1];
            m += e - b;
            for (int64_t p = b; p < e; p++) {
                const int32_t w = adj[p];
                if (levels[u] == cur && levels[w] == -1) levels[w] = cur + 1;
            }
        }
        s.levels++; s.edges_examined += m; s.frontier_total += fsize;
        fsize = vgo_gnf_level(V, levels, cur + 1, ids, parallel);   /* bfs.hpp:40-47 */
        cur++;
    }
    int64_t disc = 0;
    #pragma omp parallel for schedule(static) reduction(+ : disc) if (parallel)
    for (int32_t v = 0; v < V; v++) disc += (levels[v] > 0);
    s.discovered = disc;
    if (st) *st = s;
    free(ids);
}

/* seq_bfs.hpp:13-55: FIFO queue BFS, source level 1, unvisited -1 */
void vgo_bfs_seq(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t source, int32_t *levels)
{
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    for (int32_t v = 0; v < V; v++) levels[v] = -1;
    int64_t head = 0, tail = 0;
    levels[source] = 1; q[tail++] = source;
    while (head < tail) {
        const int32_t s = q[head++];
        for (int64_t p = rowptr[s]; p < rowptr[s + 1]; p++) {
            const int32_t v = adj[p];
            if (levels[v] == -1) { levels[v] = levels[s] + 1; q[tail++] = v; }
        }
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------ */
/* SSSP                                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* shortest_paths.hpp:85-163.  inf = numeric_limits<float>::max() - MAX_WEIGHT, which is FLT_MAX
 * in f32 (shortest_paths.hpp:102).  One iteration = save prev, relax every edge in place
 * (edge_op_push, lines 123-133), changes = #(prev != cur) (lines 143-152). */
int32_t vgo_sssp_bellman_ford(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *w,
                              int32_t source, float *dist, int parallel)
{
    const float inf_val = FLT_MAX - 100.0f;
    float *prev = (float *)malloc(sizeof(float) * (size_t)(V > 0 ? V : 1));
    #pragma omp parallel for schedule(static) if (parallel)
    for (int32_t v = 0; v < V; v++) dist[v] = inf_val;
    dist[source] = 0.0f;
    int32_t iters = 0;
    int64_t changes;
    do {
        iters++;
        #pragma omp parallel for schedule(static) if (parallel)
        for (int32_t v = 0; v < V; v++) prev[v] = dist[v];
        #pragma omp parallel for schedule(guided, 1024) if (parallel)
        for (int32_t u = 0; u < V; u++) {
            for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
                const float weight = w[p];
                const float src_weight = dist[u];
                const int32_t v = adj[p];
                const float cand = src_weight + weight;
                if (!parallel) {
                    if (dist[v] > cand) dist[v] = cand;
                } else {
                    /* same relaxation, made race-free for the multi-threaded run: compare-and-swap minimum (distances are
                     * non-negative, so the f32 order is the order of the bit patterns); the fixed point is the same */
                    int32_t *slot = (int32_t *)&dist[v];
                    int32_t seen = __atomic_load_n(slot, __ATOMIC_RELAXED), want;
                    memcpy(&want, &cand, sizeof(want));
                    while (want < seen && !__atomic_compare_exchange_n(slot, &seen, want, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
                }
            }
        }
        changes = 0;
        #pragma omp parallel for schedule(static) reduction(+ : changes) if (parallel)
        for (int32_t v = 0; v < V; v++) changes += (prev[v] != dist[v]);
    } while (changes);
    free(prev);
    return iters;
}

/* ---- SCC (checker of the reference: SCC::seq_tarjan, algorithms/scc/seq_scc.hpp): iterative Tarjan with an explicit stack.
 * The reference's labels are arbitrary counters (its test compares PARTITIONS, verify_results.h equal_components); the oracle
 * returns the canonical labelling comp[v] = smallest vertex id of v's strongly connected component. ---- */
void vgo_scc_tarjan(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp)
{
    int32_t *disc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    int32_t *low = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    int32_t *stk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));     /* Tarjan stack */
    int32_t *call_v = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));  /* DFS call stack: vertex */
    int64_t *call_p = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));  /* DFS call stack: next edge position */
    uint8_t *onstk = (uint8_t *)calloc((size_t)(V > 0 ? V : 1), 1);
    for (int32_t v = 0; v < V; v++) { disc[v] = -1; comp[v] = -1; }
    int32_t timer = 0, sp = 0;
    for (int32_t root = 0; root < V; root++) {
        if (disc[root] != -1) continue;
        int32_t cs = 0;
        call_v[cs] = root; call_p[cs] = rowptr[root]; cs++;
        disc[root] = low[root] = timer++; stk[sp++] = root; onstk[root] = 1;
        while (cs > 0) {
            const int32_t u = call_v[cs - 1];
            if (call_p[cs - 1] < rowptr[u + 1]) {
                const int32_t w = adj[call_p[cs - 1]++];
                if (disc[w] == -1) {
                    disc[w] = low[w] = timer++; stk[sp++] = w; onstk[w] = 1;
                    call_v[cs] = w; call_p[cs] = rowptr[w]; cs++;
                } else if (onstk[w] && disc[w] < low[u]) low[u] = disc[w];
            } else {
                cs--;
                if (cs > 0) { const int32_t parent = call_v[cs - 1]; if (low[u] < low[parent]) low[parent] = low[u]; }
                if (low[u] == disc[u]) {                       /* u is the root of an SCC: pop it, label with the smallest id */
                    int32_t first = sp, mn = u;
                    do { first--; if (stk[first] < mn) mn = stk[first]; } while (stk[first] != u);
                    for (int32_t i = first; i < sp; i++) { comp[stk[i]] = mn; onstk[stk[i]] = 0; }
                    sp = first;
                }
            }
        }
    }
    free(disc); free(low); free(stk); free(call_v); free(call_p); free(onstk);
}

/* ---- HITS (algorithms/hits/hits.hpp:103-173 seq_hits, the reference's own checker; f64 like apps/hits/hits.cpp:13):
 * auth = hub = 1; per step auth[v] = sum of hub over the incoming neighbours (adjacency order), normalised by the 2-norm,
 * then hub[v] = sum of auth over the outgoing neighbours, normalised.  Sequential `+=` chains, norm accumulated in vertex
 * order exactly like the reference loop. ---- */
void vgo_hits(int32_t V, const int64_t *out_rowptr, const int32_t *out_adj, const int64_t *in_rowptr, const int32_t *in_adj,
              int32_t steps, double *auth, double *hub)
{
    for (int32_t v = 0; v < V; v++) { auth[v] = 1; hub[v] = 1; }
    for (int32_t step = 0; step < steps; step++) {
        double norm = 0.0;
        for (int32_t v = 0; v < V; v++) {
            double p_auth = 0.0;
            for (int64_t p = in_rowptr[v]; p < in_rowptr[v + 1]; p++) p_auth += hub[in_adj[p]];
            auth[v] = p_auth;
            norm += p_auth * p_auth;
        }
        norm = sqrt(norm);
        for (int32_t v = 0; v < V; v++) auth[v] /= norm;
        norm = 0.0;
        for (int32_t v = 0; v < V; v++) {
            double p_hub = 0.0;
            for (int64_t p = out_rowptr[v]; p < out_rowptr[v + 1]; p++) p_hub += auth[out_adj[p]];
            hub[v] = p_hub;
            norm += p_hub * p_hub;
        }
        norm = sqrt(norm);
        for (int32_t v = 0; v < V; v++) hub[v] /= norm;
    }
}

/* ---- SSWP, single-source widest paths (algorithms/sswp/widest_paths.hpp:5-76): width[source] = inf_val, others 0; every
 * super-step pushes new = min(width[src], capacity) along every edge and keeps the larger value, until nothing changes.
 * Only min / max of the inputs: no rounding, the fixed point is unique. ---- */
int32_t vgo_sswp_bellman_ford(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *cap,
                              int32_t source, float *width, int parallel)
{
    const float inf_val = FLT_MAX - 100.0f;                 /* numeric_limits<float>::max() - MAX_WEIGHT (== FLT_MAX in f32) */
    float *prev = (float *)malloc(sizeof(float) * (size_t)(V > 0 ? V : 1));
    #pragma omp parallel for schedule(static) if (parallel)
    for (int32_t v = 0; v < V; v++) width[v] = 0.0f;
    width[source] = inf_val;
    int32_t iters = 0;
    int64_t changes;
    do {
        iters++;
        #pragma omp parallel for schedule(static) if (parallel)
        for (int32_t v = 0; v < V; v++) prev[v] = width[v];
        #pragma omp parallel for schedule(guided, 1024) if (parallel)
        for (int32_t u = 0; u < V; u++) {
            for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
                const float edge_width = cap[p];
                const float src_width = width[u];
                const float new_width = src_width < edge_width ? src_width : edge_width;
                const int32_t v = adj[p];
                if (width[v] < new_width) width[v] = new_width;
            }
        }
        changes = 0;
        #pragma omp parallel for schedule(static) reduction(+ : changes) if (parallel)
        for (int32_t v = 0; v < V; v++) changes += (prev[v] != width[v]);
    } while (changes);
    free(prev);
    return iters;
}

/* checker in the spirit of seq_widest_paths.hpp:5-64 (label-correcting with a priority queue): here a plain FIFO worklist --
 * the result is the same unique fixed point, computed in a different order from the function above */
void vgo_sswp_seq(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *cap, int32_t source, float *width)
{
    for (int32_t v = 0; v < V; v++) width[v] = 0.0f;
    width[source] = FLT_MAX;
    size_t qcap = 1024, head = 0, tail = 0, count = 0;
    int32_t *q = (int32_t *)malloc(qcap * sizeof(int32_t));
    uint8_t *inq = (uint8_t *)calloc((size_t)(V > 0 ? V : 1), 1);
    q[tail] = source; tail = (tail + 1) % qcap; count++; inq[source] = 1;
    while (count > 0) {
        const int32_t u = q[head]; head = (head + 1) % qcap; count--; inq[u] = 0;
        for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
            const int32_t v = adj[p];
            const float nw = width[u] < cap[p] ? width[u] : cap[p];
            if (nw > width[v]) {
                width[v] = nw;
                if (!inq[v]) {
                    if (count == qcap) {                     /* grow the ring */
                        int32_t *nq = (int32_t *)malloc(2 * qcap * sizeof(int32_t));
                        for (size_t i = 0; i < count; i++) nq[i] = q[(head + i) % qcap];
                        free(q); q = nq; head = 0; tail = count; qcap *= 2;
                    }
                    q[tail] = v; tail = (tail + 1) % qcap; count++; inq[v] = 1;
                }
            }
        }
    }
    free(q); free(inq);
}

/* seq_shortest_paths.hpp:9-68: lazy-deletion binary-heap Dijkstra keyed by (distance, vertex) */
typedef struct { float d; int32_t v; } vgo_hitem;
static inline int vgo_hless(vgo_hitem a, vgo_hitem b) { return (a.d < b.d) || (a.d == b.d && a.v < b.v); }
void vgo_sssp_dijkstra(int32_t V, const int64_t *rowptr, const int32_t *adj, const float *w,
                       int32_t source, float *dist)
{
    const float inf_val = FLT_MAX - 100.0f;
    for (int32_t v = 0; v < V; v++) dist[v] = inf_val;
    size_t cap = 1024, n = 0;
    vgo_hitem *h = (vgo_hitem *)malloc(cap * sizeof(vgo_hitem));
    dist[source] = 0.0f;
    h[n].d = 0.0f; h[n].v = source; n++;
    while (n > 0) {
        vgo_hitem top = h[0];
        vgo_hitem last = h[--n];
        size_t i = 0;
        for (;;) {                                   /* sift down */
            size_t l = 2 * i + 1, r = l + 1, m = i;
            vgo_hitem best = last;
            if (l < n && vgo_hless(h[l], best)) { m = l; best = h[l]; }
            if (r < n && vgo_hless(h[r], best)) { m = r; best = h[r]; }
            if (m == i) break;
            h[i] = h[m]; i = m;
        }
        if (n > 0) h[i] = last;
        const int32_t u = top.v;
        for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
            const int32_t v = adj[p];
            const float weight = w[p];
            if (dist[v] > dist[u] + weight) {
                dist[v] = dist[u] + weight;
                if (n == cap) { cap *= 2; h = (vgo_hitem *)realloc(h, cap * sizeof(vgo_hitem)); }
                vgo_hitem it; it.d = dist[v]; it.v = v;
                size_t j = n++;
                while (j > 0) {                      /* sift up */
                    size_t par = (j - 1) / 2;
                    if (!vgo_hless(it, h[par])) break;
                    h[j] = h[par]; j = par;
                }
                h[j] = it;
            }
        }
    }
    free(h);
}

/* ------------------------------------------------------------------------------------------ */
/* PageRank                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* pr.hpp:31-35,47-55,62-65 / seq_pr.hpp:32-58: in-degree minus self-loops */
void vgo_indegree_noloops(int32_t V, int64_t E, const int64_t *rowptr, const int32_t *adj, int32_t *indeg)
{
    (void)E;
    memset(indeg, 0, sizeof(int32_t) * (size_t)V);
    for (int32_t u = 0; u < V; u++)
        for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++)
            if (adj[p] != u) indeg[adj[p]]++;
}

/* pr.hpp:37-136 / seq_pr.hpp:17-96, _T = float (apps/pr/pr.cpp:28).  Expression-for-expression:
 *   d = 0.85f; k = float((1.0 - double(d)) / double(float(V))); init = float(1.0 / V);
 *   rdeg = float(1.0 / int) or 0; dangling += old / V (float / int);
 *   rank[src] += old[dst] * rdeg[dst] in adjacency order, skipping self loops;
 *   rank[src] = k + d * (rank[src] + dangling). */
void vgo_pagerank(int32_t V, const int64_t *rowptr, const int32_t *adj, const int32_t *indeg_noloops,
                  int iterations, int dangling_mode, float *ranks, int parallel)
{
    const float d = 0.85f;
    const float k = (float)((1.0 - (double)d) / (double)((float)V));
    float *old = (float *)malloc(sizeof(float) * (size_t)(V > 0 ? V : 1));
    float *rdeg = (float *)malloc(sizeof(float) * (size_t)(V > 0 ? V : 1));
    for (int32_t v = 0; v < V; v++) {
        ranks[v] = (float)(1.0 / V);
        rdeg[v] = (indeg_noloops[v] == 0) ? 0.0f : (float)(1.0 / indeg_noloops[v]);
    }
    for (int it = 0; it < iterations; it++) {
        #pragma omp parallel for schedule(static) if (parallel)
        for (int32_t v = 0; v < V; v++) { old[v] = ranks[v]; ranks[v] = 0.0f; }
        float dangling;
        if (dangling_mode == 0) {
            float acc = 0.0f;
            if (parallel) {
                #pragma omp parallel for schedule(static) reduction(+ : acc)
                for (int32_t v = 0; v < V; v++) if (indeg_noloops[v] == 0) acc += old[v] / V;
            } else {
                for (int32_t v = 0; v < V; v++) if (indeg_noloops[v] <= 0) acc += old[v] / V;
            }
            dangling = acc;
        } else {
            double acc = 0.0;
            #pragma omp parallel for schedule(static) reduction(+ : acc) if (parallel)
            for (int32_t v = 0; v < V; v++) if (indeg_noloops[v] == 0) acc += (double)(old[v] / V);
            dangling = (float)acc;
        }
        #pragma omp parallel for schedule(guided, 1024) if (parallel)
        for (int32_t u = 0; u < V; u++) {
            float acc = 0.0f;
            for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
                const int32_t v = adj[p];
                const float dst_rank = old[v];
                const float rev = rdeg[v];
                if (u != v) acc += dst_rank * rev;
            }
            ranks[u] = k + d * (acc + dangling);
        }
    }
    free(old); free(rdeg);
}

/* ------------------------------------------------------------------------------------------ */
/* Connected components                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* shiloach_vishkin.hpp:7-88: comp[v]=v; repeat { hook over every edge: if comp[src] < comp[dst]
 * then comp[dst] = comp[src]; pointer-jump comp[v] = comp[comp[v]] to a fixed point } until the
 * hook changes nothing.  Unique fixed point: comp[v] = min{u : u reaches v}. */
int32_t vgo_cc_sv(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp, int parallel)
{
    #pragma omp parallel for schedule(static) if (parallel)
    for (int32_t v = 0; v < V; v++) comp[v] = v;
    int32_t passes = 0;
    int64_t hook_changes = 1;
    while (hook_changes) {
        hook_changes = 0;
        #pragma omp parallel for schedule(guided, 1024) reduction(+ : hook_changes) if (parallel)
        for (int32_t u = 0; u < V; u++) {
            for (int64_t p = rowptr[u]; p < rowptr[u + 1]; p++) {
                const int32_t v = adj[p];
                const int32_t sv = comp[u], dv = comp[v];
                if (sv < dv) { comp[v] = sv; hook_changes++; }
            }
        }
        int64_t jump_changes = 1;
        while (jump_changes) {
            jump_changes = 0;
            #pragma omp parallel for schedule(static) reduction(+ : jump_changes) if (parallel)
            for (int32_t v = 0; v < V; v++) {
                const int32_t sv = comp[v];
                const int32_t ssv = comp[sv];
                if (sv != ssv) { comp[v] = ssv; jump_changes++; }
            }
        }
        passes++;
    }
    return passes;
}

/* seq_bfs_based.hpp:6-56: label components 1,2,3.. in order of their smallest vertex by BFS over
 * outgoing edges (COMPONENT_UNSET = -1, FIRST_COMPONENT = 1: algorithms/cc/cc.h) */
void vgo_cc_seq_bfs(int32_t V, const int64_t *rowptr, const int32_t *adj, int32_t *comp)
{
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    for (int32_t v = 0; v < V; v++) comp[v] = -1;
    int32_t cur = 1;
    for (int32_t s0 = 0; s0 < V; s0++) {
        if (comp[s0] != -1) continue;
        int64_t head = 0, tail = 0;
        comp[s0] = cur; q[tail++] = s0;
        while (head < tail) {
            const int32_t s = q[head++];
            for (int64_t p = rowptr[s]; p < rowptr[s + 1]; p++) {
                const int32_t v = adj[p];
                if (comp[v] == -1) { comp[v] = cur; q[tail++] = v; }
            }
        }
        cur++;
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------ */

uint64_t vgo_fnv1a64(const void *data, int64_t nbytes)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int64_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

int vgo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void vgo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
