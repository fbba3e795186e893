// bfs_sharded.hip -- breadth-first search over edge-cut shards behind the C ABI (one process per GPU).  The operators are those of
// algorithms/bfs/bfs.hpp:12-49, the switch rule gpu_change_state (change_state.hpp:100-141, ALPHA 15 / BETA 18) evaluated by every
// rank on the same all-gathered counters; the partition is the reference's contiguous row ranges (vect_csr/get_api.hpp:66-94).
//
// What lives where (V vertices, rank r owns the 64-aligned rows [b_r, b_r+1)):
//   levels            V entries per rank, AUTHORITATIVE on the owned rows only (elsewhere: this rank's private "already reported" marks)
//   visited bitmap    authoritative on the owned words only -- nobody rewrites V bits per level
//   frontier bitmap   replicated (bottom-up probes test arbitrary in-neighbours): it is what the exchange of a level produces
// Per level, all on the context's stream, ONE host wait (the 2 P counters that ride behind the bitmap in the same RCCL group):
//   bottom-up  probe + deferred pass over the owned unvisited rows -> owned words of the next frontier; owned visited |= them;
//              all-gather of the owned slices (V/8 bytes in total)
//   top-down   owned frontier rows expand into a candidate bitmap (any vertex); all-to-all of the candidate slices, the owner ORs
//              what it received, masks with its visited words, writes levels; all-gather of the owned slices (2 V/8 bytes per rank)
//              only while a bottom-up level may follow (before the bottom-up phase of a direction-optimising run): V/8 otherwise
//   bottom-up levels with little left to find: the found vertices travel as id lists, every rank marks them in its copy
//   small top-down levels (out-degree sum of the frontier <= half the lists' capacity; capacity per rank max(4096, V / 128 P) ids):
//              the candidates travel as id lists instead of bitmaps; the owner resolves
//              its ids exactly, the others only mark them in their frontier copy -- a superset by already-visited vertices, which no
//              probe can tell from the exact set (an unvisited vertex has no in-neighbour visited before the current level).
#include "vgl_comm.h"
#include <cstdlib>

constexpr int VGL_SHARD_NB = 512;              // workgroups of the owned-range passes (their partial counters are folded by the last one)

// owned words: candidates (OR of `parts` inputs) minus visited = the owned part of the next frontier.  WALK: the new vertices get
// their level and their out-degrees are summed (top-down levels); bottom-up levels wrote the levels themselves and need no degree sum.
template <bool WALK>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_shard_resolve(int64_t w0, int64_t nw, int parts, const uint64_t *in, int64_t part_stride, int64_t in_off,
                                                                 uint64_t *visited, uint64_t *front_new, int32_t *levels, int32_t level,
                                                                 const int64_t *rowptr, int32_t row_begin, int32_t row_end, int64_t *partials,
                                                                 uint32_t *ticket, int64_t *counts_out)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t cnt = 0, deg = 0;
    const int lane = vgl_lane();
    const int64_t nround = (nw + 63) & ~(int64_t)63;
    for (int64_t wi = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; wi < nround; wi += (int64_t)gridDim.x * VGL_BLOCK) {
        uint64_t w = 0;
        if (wi < nw) {
            for (int p = 0; p < parts; p++) w |= in[(int64_t)p * part_stride + in_off + wi];
            const uint64_t vis = visited[w0 + wi];
            w &= ~vis;
            if (w) visited[w0 + wi] = vis | w;
            front_new[w0 + wi] = w;
            cnt += __popcll(w);
        }
        if (WALK) {
            unsigned long long todo = __ballot(w != 0);
            while (todo) {                                       // the wavefront walks its non-zero words together: one coalesced row of levels each
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint64_t ww = __shfl(w, l);
                const int64_t v = ((w0 + wi - lane + l) << 6) + lane;
                if (((ww >> lane) & 1ULL) && v < row_end) {
                    levels[v] = level;
                    const int64_t r = v - row_begin;
                    deg += rowptr[r + 1] - rowptr[r];
                }
            }
        }
    }
    cnt = vgl_block_reduce_add(cnt, s64);
    deg = vgl_block_reduce_add(deg, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(partials + 2 * blockIdx.x, cnt) ^ vgl_put_agent(partials + 2 * blockIdx.x + 1, deg);
    if (!vgl_last_block(ticket, dep)) return;
    int64_t a = 0, b = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += VGL_BLOCK) { a += vgl_load_agent(partials + 2 * i); b += vgl_load_agent(partials + 2 * i + 1); }
    a = vgl_block_reduce_add(a, s64);
    b = vgl_block_reduce_add(b, s64);
    if (threadIdx.x == 0) { counts_out[0] = a; counts_out[1] = b; counts_out[2] = 0; counts_out[3] = 0; }
}

// `parts` id lists (stride 1 + cap): owned ids are claimed on the visited bitmap (a vertex reported by several ranks is taken once), get
// their level and count; every listed id is marked in the frontier copy.  A list that overflowed (count > cap) voids the level's sparse
// exchange for everybody: counts_out[2] = 1 and nothing is touched.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_shard_apply_ids(int parts, int32_t cap, const int32_t *lists, int32_t V, int32_t row_begin, int32_t row_end,
                                                                   uint64_t *visited, uint64_t *front_new, int32_t *levels, int32_t level,
                                                                   const int64_t *rowptr, int64_t *partials, uint32_t *ticket, int64_t *counts_out)
{
    __shared__ int64_t s64[VGL_WAVES];
    bool overflow = false;
    for (int p = 0; p < parts; p++) overflow |= lists[(int64_t)p * (1 + cap)] > cap;      // the same answer in every workgroup
    int64_t cnt = 0, deg = 0;
    for (int p = 0; p < (overflow ? 0 : parts); p++) {
        const int32_t *list = lists + (int64_t)p * (1 + cap);
        const int32_t n = list[0];
        for (int32_t i = (int32_t)(blockIdx.x * VGL_BLOCK + threadIdx.x); i < n; i += (int32_t)(gridDim.x * VGL_BLOCK)) {
            const int32_t v = list[1 + i];
            if (v < 0 || v >= V) continue;
            const unsigned long long bit = 1ULL << (v & 63);
            if (v >= row_begin && v < row_end) {
                const unsigned long long old = atomicOr((unsigned long long *)&visited[v >> 6], bit);
                if (old & bit) continue;
                levels[v] = level;
                cnt++;
                deg += rowptr[v - row_begin + 1] - rowptr[v - row_begin];
            }
            atomicOr((unsigned long long *)&front_new[v >> 6], bit);
        }
    }
    cnt = vgl_block_reduce_add(cnt, s64);
    deg = vgl_block_reduce_add(deg, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(partials + 2 * blockIdx.x, cnt) ^ vgl_put_agent(partials + 2 * blockIdx.x + 1, deg);
    if (!vgl_last_block(ticket, dep)) return;
    int64_t a = 0, b = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += VGL_BLOCK) { a += vgl_load_agent(partials + 2 * i); b += vgl_load_agent(partials + 2 * i + 1); }
    a = vgl_block_reduce_add(a, s64);
    b = vgl_block_reduce_add(b, s64);
    if (threadIdx.x == 0) { counts_out[0] = a; counts_out[1] = b; counts_out[2] = overflow ? 1 : 0; counts_out[3] = 0; }
}

// Bottom-up levels that find little: the owners have resolved their rows already (probe + vgl_k_shard_resolve<false>); the found
// vertices travel as id lists and every rank marks the OTHER ranks' ids in its copy of the next frontier (its own words are in place).
// counts_all[4 p] = rank p's number of found vertices, [4 p + 2] = 1 when any list overflowed (then nothing is marked and the
// caller falls back to the bitmap exchange; the same answer on every rank).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_shard_mark_ids(int parts, int self, int32_t cap, const int32_t *lists, int32_t V, uint64_t *front_new,
                                                                  int64_t *counts_all)
{
    bool overflow = false;
    for (int p = 0; p < parts; p++) overflow |= lists[(int64_t)p * (1 + cap)] > cap;
    if (blockIdx.x == 0)
        for (int p = threadIdx.x; p < parts; p += VGL_BLOCK) {
            counts_all[4 * p] = lists[(int64_t)p * (1 + cap)];
            counts_all[4 * p + 1] = 0; counts_all[4 * p + 2] = overflow ? 1 : 0; counts_all[4 * p + 3] = 0;
        }
    if (overflow) return;
    for (int p = 0; p < parts; p++) {
        if (p == self) continue;
        const int32_t *list = lists + (int64_t)p * (1 + cap);
        const int32_t n = list[0];
        for (int32_t i = (int32_t)(blockIdx.x * VGL_BLOCK + threadIdx.x); i < n; i += (int32_t)(gridDim.x * VGL_BLOCK)) {
            const int32_t v = list[1 + i];
            if (v >= 0 && v < V) atomicOr((unsigned long long *)&front_new[v >> 6], 1ULL << (v & 63));
        }
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_shard_bfs_init(int64_t words, int32_t source, uint64_t *visited, uint64_t *front, uint64_t *front_new,
                                                                  uint32_t *ticket)
{
    for (int64_t w = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; w < words; w += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t bit = (w == (source >> 6)) ? (1ULL << (source & 63)) : 0ULL;
        visited[w] = bit; front[w] = bit; front_new[w] = 0;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < VGL_TICKET_WORDS; i += VGL_BLOCK) ticket[i] = 0;
}

extern "C" int vgl_hip_bfs_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *given, vgl_hip_graph *g, int32_t source, int mode, int64_t global_edges,
                                       int gather_levels, int32_t *d_levels, vgl_hip_bfs_stats *stats)
{
    if (!c || !g || !d_levels) VGL_FAIL("bfs_run_sharded: null argument");
    if (mode != VGL_HIP_BFS_TOP_DOWN && mode != VGL_HIP_BFS_DIRECTION_OPT) VGL_FAIL("bfs_run_sharded: unknown mode");
    if (mode == VGL_HIP_BFS_DIRECTION_OPT && !g->in.rowptr) VGL_FAIL("bfs_run_sharded: direction-optimising mode needs the incoming CSR of the owned rows");
    const int32_t V = g->V;
    if (source < 0 || source >= V) VGL_FAIL("bfs_run_sharded: source vertex out of range");
    if ((g->row_begin & 63) || (g->row_end != V && (g->row_end & 63))) VGL_FAIL("bfs_run_sharded: the owned row range must start and end on multiples of 64");
    // (the solo communicator of sharded.hip, inlined: a world of one needs only the hand-over buffers)
    vgl_hip_comm local;
    vgl_hip_comm *m = given;
    if (!m) {
        m = &local;
        m->ctx = c;
        VGL_HIP_TRY(hipMalloc((void **)&m->d_small, sizeof(int64_t) * VGL_COMM_SMALL));
        VGL_HIP_TRY(hipHostMalloc((void **)&m->h_small, sizeof(int64_t) * (VGL_COMM_SMALL + 8), hipHostMallocDefault));
        for (int i = 0; i < VGL_COMM_SMALL + 8; i++) m->h_small[i] = 0;
    }
    struct cleanup {
        vgl_hip_comm *l; bool on;
        ~cleanup()
        {
            if (!on) return;
            hipStreamSynchronize(l->ctx->stream);
            for (int i = 0; i < VGL_COMM_SCRATCH_SLOTS; i++) if (l->scratch[i]) hipFree(l->scratch[i]);
            hipFree(l->d_small); hipHostFree(l->h_small);
        }
    } guard{&local, given == nullptr};
    if (m->ctx != c) VGL_FAIL("bfs_run_sharded: the communicator belongs to another context");
    const int P = m->world, rank = m->rank;
    if (P == 1 && (g->row_begin != 0 || g->row_end != V)) VGL_FAIL("bfs_run_sharded: a world of one must own all rows");
    if (P > 64) VGL_FAIL("bfs_run_sharded: at most 64 ranks");
    m->stats = {0, 0, 0, 0, 0, 0};
    const int64_t words = vgl_ceil_div(V, 64);
    const int64_t w0 = g->row_begin >> 6, nw = vgl_ceil_div(g->row_end, 64) - w0;
    const int64_t *bounds = nullptr;
    std::vector<int64_t> word_bb((size_t)P + 1, 0), level_bb((size_t)P + 1, 0);
    bool equal = true;
    const bool active = vgl_comm_active(m);
    if (active) {
        VGL_TRY(vgl_comm_row_bounds(m, g, &bounds));
        for (int p = 0; p <= P; p++) {
            if (p < P && (bounds[p] & 63)) VGL_FAIL("bfs_run_sharded: every rank's first row must be a multiple of 64");
            word_bb[(size_t)p] = (p == P ? words : bounds[p] / 64) * 8;
            level_bb[(size_t)p] = bounds[p] * 4;
        }
        for (int p = 0; p < P; p++) equal = equal && (word_bb[(size_t)p + 1] - word_bb[(size_t)p]) == (int64_t)(words / P) * 8;
        equal = equal && words % P == 0;
    }
    // capacity of a rank's id list: lists pay while all of them together stay well below the V/8 bytes of a bitmap exchange
    int32_t sparse_cap = (int32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(4096, (int64_t)V / (128 * (int64_t)P)));
    vgl_ctx_refresh_env(c);
    if (c->bfs.shard_sparse_cap >= 0) sparse_cap = c->bfs.shard_sparse_cap;
    if (!active) sparse_cap = 0;
    const int64_t nz_total = active ? bounds[(size_t)P + 1] : 0;         // rows with incoming edges, all ranks: what a bottom-up level can find at most

    // one block of scratch, carved: visited | front A | front B | candidates | tickets | partials | counts (mine, all) | id lists
    const size_t bm = sizeof(uint64_t) * (size_t)(words + 1);
    const size_t lists_bytes = sizeof(int32_t) * (size_t)(1 + sparse_cap) * (size_t)(P + 1);
    const size_t fixed = sizeof(uint32_t) * VGL_TICKET_WORDS + sizeof(int64_t) * (2 * VGL_SHARD_NB + 4 + 4 * (size_t)P) + 256;
    char *base = nullptr;
    VGL_TRY(vgl_comm_scratch(m, 3, 4 * bm + fixed + lists_bytes + 64, (void **)&base));
    uint64_t *visited = reinterpret_cast<uint64_t *>(base), *front = visited + (words + 1), *front_new = front + (words + 1), *cand = front_new + (words + 1);
    int64_t *partials = reinterpret_cast<int64_t *>(cand + (words + 1));
    int64_t *my_counts = partials + 2 * VGL_SHARD_NB, *all_counts = my_counts + 4;
    uint32_t *ticket = reinterpret_cast<uint32_t *>(all_counts + 4 * (size_t)P);
    int32_t *my_list = reinterpret_cast<int32_t *>(ticket + VGL_TICKET_WORDS + 16), *all_lists = my_list + (1 + sparse_cap);
    uint64_t *recv = nullptr;
    if (active) VGL_TRY(vgl_comm_scratch(m, 4, sizeof(uint64_t) * (size_t)(equal ? words : words * P), (void **)&recv));

    VGL_TRY(vgl_hip_bfs_init(c, V, source, d_levels));
    hipLaunchKernelGGL(vgl_k_shard_bfs_init, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, vgl_ceil_div(words, VGL_BLOCK)))), dim3(VGL_BLOCK), 0,
                       c->stream, words, source, visited, front, front_new, ticket);
    VGL_HIP_TRY(hipGetLastError());

    vgl_hip_bfs_stats st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool direction_opt = mode == VGL_HIP_BFS_DIRECTION_OPT;
    const int64_t E = global_edges > 0 ? global_edges : g->out.edges;
    const int64_t factor = std::max<int64_t>(1, (E / V) / 2);                   // change_state.hpp:104
    int64_t F = 1, M = 0, prevF = 0, visited_total = 0;
    bool bottom_up = false;
    // Who reads the words of the frontier a rank does not own: only a bottom-up level (its probes test arbitrary in-neighbours); a
    // top-down level expands owned rows.  So a top-down level hands its next frontier round only while a bottom-up level can still
    // follow it without notice -- in a direction-optimising run before the bottom-up phase; once that phase is over (and in a pure
    // top-down run) the V/8-byte all-gather is left out, and a level that turns bottom-up after all fetches the slices first.
    bool been_bottom_up = false, replicated = true;
    const unsigned nb_own = (unsigned)std::max<int64_t>(1, std::min<int64_t>(VGL_SHARD_NB, vgl_ceil_div(nw, VGL_BLOCK)));
    int64_t h_counts[4 * 64];
    for (int32_t level = 1;; level++) {
        visited_total += F;
        st.levels++; st.frontier_total += F;
        if (direction_opt) {                                                    // gpu_change_state on the frontier about to be expanded
            if (!bottom_up) {
                if (F > prevF && M >= ((V - visited_total) * factor + V) / 15) bottom_up = true;
            } else if (F <= prevF && F < ((V - visited_total) * factor + V) / (factor * 18)) bottom_up = false;
        }
        prevF = F;
        bool exchanged_sparse = false;
        if (bottom_up && !replicated) {
            if (active) VGL_TRY(vgl_comm_allgatherv_inplace(m, front, word_bb.data()));
            replicated = true;
        }
        been_bottom_up = been_bottom_up || bottom_up;
        if (bottom_up) {
            // owned unvisited rows look for a parent in the replicated frontier; the owned words of front_new = what they found
            int64_t found = 0, probed = 0;
            VGL_TRY(vgl_hip_bfs_step_bottom_up(c, g, d_levels, level, visited, front, front_new, stats ? &found : nullptr, stats ? &probed : nullptr));
            st.bu_steps++; st.bu_edges += probed; st.bu_found += found; st.edges_examined += probed;
            {
                vgl_timed_launch tl(c, "bfs_shard_resolve");
                hipLaunchKernelGGL(vgl_k_shard_resolve<false>, dim3(nb_own), dim3(VGL_BLOCK), 0, c->stream, w0, nw, 1, (const uint64_t *)front_new, (int64_t)0, w0,
                                   visited, front_new, d_levels, level + 1, g->out.rowptr, g->row_begin, g->row_end, partials, ticket, my_counts);
            }
            // late bottom-up levels find little (RMAT-27: 4e7, 6e5, 2e3 vertices on the three levels): when at most 4 lists' worth of
            // rows with incoming edges is still unvisited the found vertices travel as ids; an overflow falls back to the bitmaps
            if (sparse_cap > 0 && nz_total > 0 && nz_total - visited_total <= 4 * (int64_t)sparse_cap * P) {
                VGL_TRY(vgl_bitmap_to_ids(c, nw, front_new + w0, w0, sparse_cap, my_list));
                VGL_TRY(vgl_comm_allgather(m, my_list, all_lists, sizeof(int32_t) * (size_t)(1 + sparse_cap)));
                {
                    vgl_timed_launch tl(c, "bfs_shard_resolve");
                    hipLaunchKernelGGL(vgl_k_shard_mark_ids, dim3(128), dim3(VGL_BLOCK), 0, c->stream, P, rank, sparse_cap, (const int32_t *)all_lists, V, front_new,
                                       all_counts);
                }
                VGL_HIP_TRY(hipGetLastError());
                VGL_TRY(vgl_comm_read_small(m, all_counts, 4 * P, h_counts));
                if (h_counts[2] == 0) { exchanged_sparse = true; m->stats.sparse_levels++; }
            }
        } else {
            int64_t Fl = 0, Ml = 0;
            VGL_TRY(vgl_hip_bfs_step_top_down_bits(c, g, d_levels, level, visited, front, cand, &Fl, &Ml));
            st.td_steps++; st.td_frontier += Fl; st.td_edges += Ml; st.edges_examined += Ml;
            // id lists when the level cannot produce many candidates: M (the out-degree sum of the frontier, known after a top-down
            // level) bounds them; after a bottom-up level and for the source only the frontier size is known
            if (sparse_cap > 0 && (M > 0 ? M <= (int64_t)sparse_cap * P / 2 : F <= sparse_cap)) {
                VGL_TRY(vgl_hip_bitmap_to_ids(c, words, cand, sparse_cap, my_list));
                VGL_TRY(vgl_comm_allgather(m, my_list, all_lists, sizeof(int32_t) * (size_t)(1 + sparse_cap)));
                VGL_TRY(vgl_zero_words(c, front_new, words));
                {
                vgl_timed_launch tl(c, "bfs_shard_resolve");
                hipLaunchKernelGGL(vgl_k_shard_apply_ids, dim3(128), dim3(VGL_BLOCK), 0, c->stream, P, sparse_cap, (const int32_t *)all_lists, V, g->row_begin,
                                   g->row_end, visited, front_new, d_levels, level + 1, g->out.rowptr, partials, ticket, my_counts);
                }
                VGL_TRY(vgl_comm_allgather(m, my_counts, all_counts, sizeof(int64_t) * 4));
                VGL_TRY(vgl_comm_read_small(m, all_counts, 4 * P, h_counts));
                if (h_counts[2] == 0) { exchanged_sparse = true; m->stats.sparse_levels++; }      // (every rank computed the same overflow flag)
            }
            if (!exchanged_sparse) {
                const uint64_t *in = cand;
                int64_t stride = 0, off = w0;
                if (active) {
                    if (equal) { VGL_TRY(vgl_comm_alltoall(m, cand, recv, (words / P) * 8)); in = recv; stride = words / P; off = 0; }
                    else { VGL_TRY(vgl_comm_allgather(m, cand, recv, words * 8)); in = recv; stride = words; off = w0; }
                }
                vgl_timed_launch tl(c, "bfs_shard_resolve");
                hipLaunchKernelGGL(vgl_k_shard_resolve<true>, dim3(nb_own), dim3(VGL_BLOCK), 0, c->stream, w0, nw, P, in, stride, off, visited, front_new,
                                   d_levels, level + 1, g->out.rowptr, g->row_begin, g->row_end, partials, ticket, my_counts);
            }
        }
        VGL_HIP_TRY(hipGetLastError());
        if (exchanged_sparse) replicated = true;                                // (a superset by vertices of earlier levels, see the header)
        else {
            // the owned slices of the next frontier and the two counters of every rank, one fused RCCL launch
            const bool hand_round = direction_opt && (bottom_up || !been_bottom_up);
            replicated = hand_round || !active;
            vgl_comm_group_begin(m);
            {
                // (a failing collective must not leave the group open: every later collective would queue behind it for ever)
                int rc = 0;
                if (active && hand_round) rc = vgl_comm_allgatherv_inplace(m, front_new, word_bb.data());
                if (!rc) rc = vgl_comm_allgather(m, my_counts, all_counts, sizeof(int64_t) * 4);
                const int rc_end = vgl_comm_group_end(m);
                if (rc || rc_end) return 1;
            }
            VGL_TRY(vgl_comm_read_small(m, all_counts, 4 * P, h_counts));
        }
        F = 0; M = 0;
        for (int p = 0; p < P; p++) { F += h_counts[4 * p]; M += h_counts[4 * p + 1]; }
        std::swap(front, front_new);
        if (F == 0) break;
    }
    (void)rank;
    if (gather_levels && active) VGL_TRY(vgl_comm_allgatherv_inplace(m, d_levels, level_bb.data()));
    st.discovered = visited_total;
    st.algorithmic_bytes = 8 * st.edges_examined + 20 * st.td_frontier + 4 * (int64_t)(g->row_end - g->row_begin) + (int64_t)st.bu_steps * (nw * 8);
    if (stats) *stats = st;
    return 0;
}
