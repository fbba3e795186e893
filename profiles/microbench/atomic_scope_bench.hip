// atomic_scope_bench.hip -- what it costs to SET random bits of a 2 MiB bitmap (the next-frontier bitmap of a top-down BFS level over
// 16.8 M vertices) from every CU of an MI355X, by the form of the update:
//   agent-scope atomicOr (what vgl_k_td_expand<EMIT> does: executed at the memory side, one request per lane)
//   workgroup-scope atomicOr into a PER-XCD copy of the bitmap (index = HW_REG_XCC_ID): only workgroups of one XCD touch a copy, the XCD's
//     L2 is their coherence point, so the narrower scope is sufficient for them; a later pass ORs the 8 copies
//   plain 4-byte stores into a 64 MiB array (levels[dst] = level), plain byte stores into a 16 MiB byte map
//   random 8-byte loads of the bitmap (the cost of asking first)
// and checks that the OR of the per-XCD copies equals the agent-scope result.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_scope_bench atomic_scope_bench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }      // HW_REG_XCC_ID[3:0]

constexpr int PER = 8;
enum { AGENT_OR64, WG_OR64_XCD, WG_OR32_XCD, AGENT_OR64_RET, WG_OR64_XCD_RET, STORE32, STORE8, LOAD64, AGENT_OR32 };

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *targets, int64_t n, uint64_t *bits, int64_t words, int32_t *levels, uint8_t *bytes, uint64_t *sink_out,
                                         float dup)
{
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint64_t *mine = bits;
    if (MODE == WG_OR64_XCD || MODE == WG_OR32_XCD || MODE == WG_OR64_XCD_RET) mine = bits + (int64_t)xcc_id() * words;
    uint32_t t[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) { const int64_t i = base + (int64_t)j * stride; t[j] = i < n ? targets[i] : 0xFFFFFFFFu; }
    uint64_t sink = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if (t[j] == 0xFFFFFFFFu) continue;
        const uint32_t v = t[j];
        if (MODE == AGENT_OR64) __hip_atomic_fetch_or(&mine[v >> 6], 1ULL << (v & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == AGENT_OR32) __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(mine) + (v >> 5), 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == WG_OR64_XCD) __hip_atomic_fetch_or(&mine[v >> 6], 1ULL << (v & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == WG_OR32_XCD) __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(mine) + (v >> 5), 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == AGENT_OR64_RET) sink += __hip_atomic_fetch_or(&mine[v >> 6], 1ULL << (v & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == WG_OR64_XCD_RET) sink += __hip_atomic_fetch_or(&mine[v >> 6], 1ULL << (v & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == STORE32) levels[v] = 7;
        else if (MODE == STORE8) bytes[v] = 1;
        else if (MODE == LOAD64) sink += mine[v >> 6];
    }
    if (sink == 0x123456789ULL) sink_out[0] = sink;
}

__global__ void k_or_copies(uint64_t *bits, int64_t words, uint64_t *out)
{
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) {
        uint64_t a = 0;
        for (int x = 0; x < 8; x++) a |= bits[(int64_t)x * words + w];
        out[w] = a;
    }
}

struct bufs { uint32_t *targets; uint64_t *bits, *ref, *merged, *sink; int32_t *levels; uint8_t *bytes; int64_t words, n; };

template <int MODE> double run(const bufs &b, const char *what, int64_t n)
{
    hipMemset(b.bits, 0, (size_t)b.words * 8 * 8);
    const int blocks = (int)((n + 256 * PER - 1) / (256 * PER));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, b.targets, n, b.bits, b.words, b.levels, b.bytes, b.sink, 0.f);
    hipMemset(b.bits, 0, (size_t)b.words * 8 * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, b.targets, n, b.bits, b.words, b.levels, b.bytes, b.sink, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s n = %9lld  %8.1f us  %7.1f G updates/s\n", what, (long long)n, ms * 1e3, (double)n / ms / 1e6);
    return ms;
}

int main()
{
    const int64_t V = 1LL << 24, words = V / 64;
    bufs b;
    b.words = words;
    const int64_t NMAX = 1LL << 24;
    hipMalloc(&b.targets, NMAX * 4);
    hipMalloc(&b.bits, words * 8 * 8);
    hipMalloc(&b.ref, words * 8);
    hipMalloc(&b.merged, words * 8);
    hipMalloc(&b.sink, 64);
    hipMalloc(&b.levels, V * 4);
    hipMalloc(&b.bytes, V);
    hipMemset(b.levels, 0xFF, V * 4);
    hipMemset(b.bytes, 0, V);
    // targets: (a) uniform over all vertices; (b) RMAT-like skew (most targets among the low ids, many repeats)
    for (int skew = 0; skew < 2; skew++) {
        std::vector<uint32_t> h(NMAX);
        uint64_t x = 88172645463325252ULL;
        for (int64_t i = 0; i < NMAX; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t v = (uint32_t)(x >> 20) & (uint32_t)(V - 1);
            if (skew) { const int sh = (int)((x >> 3) % 12); v >>= sh; }      // geometric mix of ranges: half of the targets below 2^18
            h[i] = v;
        }
        hipMemcpy(b.targets, h.data(), NMAX * 4, hipMemcpyHostToDevice);
        printf("---- targets: %s ----\n", skew ? "skewed (geometric mix of id ranges, many repeats)" : "uniform over 2^24 vertices");
        for (int64_t n : {1LL << 20, 1LL << 22, 1LL << 24}) {
            run<LOAD64>(b, "random 8-byte loads of the 2 MiB bitmap", n);
            run<STORE32>(b, "plain 4-byte stores into 64 MiB (levels)", n);
            run<STORE8>(b, "plain byte stores into a 16 MiB byte map", n);
            run<AGENT_OR64>(b, "agent-scope atomicOr u64, no return", n);
            hipMemcpy(b.ref, b.bits, words * 8, hipMemcpyDeviceToDevice);
            run<AGENT_OR32>(b, "agent-scope atomicOr u32, no return", n);
            run<AGENT_OR64_RET>(b, "agent-scope atomicOr u64, value returned", n);
            run<WG_OR64_XCD>(b, "workgroup-scope atomicOr u64, per-XCD copy", n);
            hipLaunchKernelGGL(k_or_copies, dim3(1024), dim3(256), 0, 0, b.bits, words, b.merged);
            std::vector<uint64_t> r(words), m(words);
            hipMemcpy(r.data(), b.ref, words * 8, hipMemcpyDeviceToHost);
            hipMemcpy(m.data(), b.merged, words * 8, hipMemcpyDeviceToHost);
            int64_t bad = 0, set = 0;
            for (int64_t w = 0; w < words; w++) { bad += r[w] != m[w]; set += __builtin_popcountll(r[w]); }
            printf("    OR of the 8 per-XCD copies vs agent-scope bitmap: %lld words differ (%lld bits set)\n", (long long)bad, (long long)set);
            run<WG_OR32_XCD>(b, "workgroup-scope atomicOr u32, per-XCD copy", n);
            run<WG_OR64_XCD_RET>(b, "workgroup-scope atomicOr u64 per-XCD, value returned", n);
        }
    }
    // census of XCC ids seen by the blocks of one launch
    hipDeviceSynchronize();
    return 0;
}
