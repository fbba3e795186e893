// cc.hip -- Shiloach-Vishkin connected components (CC::vgl_shiloach_vishkin, algorithms/cc/shiloach_vishkin.hpp:7-88).
//
// hook : for every edge (src,dst): if comp[src] < comp[dst] then comp[dst] = comp[src]   (edge_op, lines 37-49); done with an
//        integer atomic-min so no update is lost.  Edge-balanced like the SSSP relax kernel: workgroup = 2048 CSR edges.
// jump : comp[v] = comp[comp[v]] repeated to the root (jump_op loop, lines 56-76) in a single pass per vertex.
// The loop ends when a hook pass changes nothing; the fixed point comp[v] = min{u : u reaches v} is unique, so labels are
// bit-identical to the reference under identity (CSR) numbering.  Algorithmic bytes: 8*E + 12*V per hook, 12*V per jump.
#include "vgl_hip_internal.h"

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_init(int32_t V, int32_t *comp)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) comp[v] = v;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_hook(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row,
                                                           int64_t E, int32_t row_base, int32_t *comp, int64_t *counters)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
    const int i0 = threadIdx.x * VGL_EPT;
    int changed = 0;
    if (i0 < n) {
        int32_t dsts[VGL_EPT];
        if (i0 + VGL_EPT <= n) {
            const int4 a0 = *reinterpret_cast<const int4 *>(adj + e0 + i0);
            const int4 a1 = *reinterpret_cast<const int4 *>(adj + e0 + i0 + 4);
            dsts[0] = a0.x; dsts[1] = a0.y; dsts[2] = a0.z; dsts[3] = a0.w; dsts[4] = a1.x; dsts[5] = a1.y; dsts[6] = a1.z; dsts[7] = a1.w;
        } else {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) dsts[j] = (i0 + j < n) ? adj[e0 + i0 + j] : 0;
        }
        int32_t csrc[VGL_EPT], cdst[VGL_EPT];
        int prev_row = -1;
        int32_t cs = 0;
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            const int row = s_map[i0 + j];
            if (row != prev_row) { prev_row = row; cs = comp[row_base + r_first + row]; }
            csrc[j] = cs;
            cdst[j] = (i0 + j < n) ? comp[dsts[j]] : -1;
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            if (csrc[j] < cdst[j]) {
                const int32_t before = atomicMin(comp + dsts[j], csrc[j]);
                if (before > csrc[j]) changed = 1;
            }
        }
    }
    if (__syncthreads_or(changed) && threadIdx.x == 0) counters[C_CHANGED] = 1;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_jump(int32_t V, int32_t *comp)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        int32_t c = comp[v];
        int32_t cc = comp[c];
        if (c != cc) {
            while (cc != c) { c = cc; cc = comp[c]; }       // labels only decrease and comp[x] <= x: terminates at a root
            comp[v] = c;
        }
    }
}

static inline unsigned vgl_grid2(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(n, VGL_BLOCK))); }

static int vgl_cc_hook_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *comp)
{
    if (g->out.ntiles == 0) return 0;
    vgl_timed_launch tl(c, "cc_hook");
    hipLaunchKernelGGL(vgl_k_cc_hook, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj,
                       g->out.tile_row, g->out.edges, g->row_begin, comp, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_cc_init(vgl_hip_ctx *c, int32_t V, int32_t *d_comp)
{
    if (!c || !d_comp) VGL_FAIL("cc_init: null argument");
    hipLaunchKernelGGL(vgl_k_cc_init, dim3(vgl_grid2(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_cc_hook_owned(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, int *changed)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_hook_owned: null argument");
    VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
    VGL_TRY(vgl_cc_hook_launch(c, g, d_comp));
    VGL_TRY(vgl_read_counters(c));
    if (changed) *changed = (int)c->h_counters[C_CHANGED];
    return 0;
}

int vgl_hip_cc_jump(vgl_hip_ctx *c, int32_t V, int32_t *d_comp)
{
    if (!c || !d_comp) VGL_FAIL("cc_jump: null argument");
    vgl_timed_launch tl(c, "cc_jump");
    hipLaunchKernelGGL(vgl_k_cc_jump, dim3(vgl_grid2(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_cc_run(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_run: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("cc_run: graph handle must own all rows (use the step API for shards)");
    VGL_TRY(vgl_hip_cc_init(c, g->V, d_comp));
    vgl_hip_cc_stats st = {0, 0};
    for (;;) {
        int changed = 0;
        VGL_TRY(vgl_hip_cc_hook_owned(c, g, d_comp, &changed));
        st.hook_passes++;
        st.algorithmic_bytes += 8 * g->out.edges + 12 * (int64_t)g->V;
        if (!changed) break;                         // while(hook_changes)  (shiloach_vishkin.hpp:29)
        VGL_TRY(vgl_hip_cc_jump(c, g->V, d_comp));
        st.algorithmic_bytes += 12 * (int64_t)g->V;
    }
    if (stats) *stats = st;
    return 0;
}

}  // extern "C"
