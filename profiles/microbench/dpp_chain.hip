// micro-benchmark: sequential f32 sum of n values by one wavefront -- DPP wave_shr chain against the LDS-broadcast fold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float chain64(float acc_in, float v)
{
    const int lane = threadIdx.x & 63;
    float a = lane == 0 ? __fadd_rn(acc_in, v) : v;
#pragma unroll
    for (int i = 0; i < 63; i++)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(v));
    return __shfl(a, 63);
}

__global__ void k_dpp(const float *in, int n, float *out, long long *cycles)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    const long long t0 = clock64();
    for (int base = 0; base < n; base += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = base + u * 64 + lane; v[u] = q < n ? in[q] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = chain64(acc, v[u]);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { *out = acc; *cycles = t1 - t0; }
}

__global__ void k_lds(const float *in, int n, float *out, long long *cycles)
{
    __shared__ float cur[512];
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    const long long t0 = clock64();
    for (int base = 0; base < n; base += 512) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = base + u * 64 + lane; cur[u * 64 + lane] = q < n ? in[q] : 0.0f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 8
        for (int i = 0; i < 128; i++) {
            acc = __fadd_rn(acc, cur[4 * i]); acc = __fadd_rn(acc, cur[4 * i + 1]); acc = __fadd_rn(acc, cur[4 * i + 2]); acc = __fadd_rn(acc, cur[4 * i + 3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { *out = acc; *cycles = t1 - t0; }
}

// variant: the next 32 values are read from LDS into registers while the current 32 are added (explicit double buffer)
__global__ void k_lds2(const float *in, int n, float *out, long long *cycles)
{
    __shared__ float4 cur4[128];
    float *cur = reinterpret_cast<float *>(cur4);
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    const long long t0 = clock64();
    for (int base = 0; base < n; base += 512) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = base + u * 64 + lane; cur[u * 64 + lane] = q < n ? in[q] : 0.0f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = cur4[j];
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = cur4[(g + 1) * 8 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) { acc = __fadd_rn(acc, a[j].x); acc = __fadd_rn(acc, a[j].y); acc = __fadd_rn(acc, a[j].z); acc = __fadd_rn(acc, a[j].w); }
            if (g + 2 < 16) {
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = cur4[(g + 2) * 8 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) { acc = __fadd_rn(acc, b[j].x); acc = __fadd_rn(acc, b[j].y); acc = __fadd_rn(acc, b[j].z); acc = __fadd_rn(acc, b[j].w); }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { *out = acc; *cycles = t1 - t0; }
}

// variant: no LDS at all -- the 512 values of a batch stay in the wavefront's registers, the chain reads them with v_readlane
// (lane index a compile-time constant) and adds the scalar
template <int U0>
__device__ __forceinline__ float chain_regs(float acc, const float (&v)[8])
{
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
        for (int l = 0; l < 64; l++) acc = __fadd_rn(acc, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[u]), l)));
    return acc;
}
__global__ void k_readlane(const float *in, int n, float *out, long long *cycles)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    const long long t0 = clock64();
    float v[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int q = u * 64 + lane; v[u] = q < n ? in[q] : 0.0f; }
    for (int base = 0; base < n; base += 512) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = base + 512 + u * 64 + lane; w[u] = q < n ? in[q] : 0.0f; }      // next batch in flight
        acc = chain_regs<0>(acc, v);
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = w[u];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { *out = acc; *cycles = t1 - t0; }
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned x = 12345;
    for (int i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) * (1.0f / 16777216.0f) * 1e-3f; }
    float ref = 0.0f;
    for (int i = 0; i < n; i++) ref = ref + h[i];
    float *d_in, *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_in, n * 4)); CHECK(hipMalloc(&d_out, 4)); CHECK(hipMalloc(&d_cyc, 8));
    CHECK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int which = 0; which < 4; which++) {
        for (int rep = 0; rep < 2; rep++) {
            if (which == 0) hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d_in, n, d_out, d_cyc);
            else if (which == 1) hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d_in, n, d_out, d_cyc);
            else if (which == 2) hipLaunchKernelGGL(k_lds2, dim3(1), dim3(64), 0, 0, d_in, n, d_out, d_cyc);
            else hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, d_in, n, d_out, d_cyc);
            CHECK(hipDeviceSynchronize());
        }
        float got; long long cyc;
        CHECK(hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("%s: sum %.9g (sequential reference %.9g, %s)  %.2f clock64 ticks per element\n", which == 0 ? "dpp chain" : which == 1 ? "lds fold " : which == 2 ? "lds fold, double-buffered registers" : "readlane chain (no LDS)", got, ref,
               got == ref ? "bit-identical" : "DIFFERENT", (double)cyc / n);
    }
    return 0;
}
