// shadow_memory.h -- where user data lives under __USE_HIP__ (included by vgl_runtime/helpers/memory_API/memory_API.h).
//
// The CUDA flavour of the reference puts everything into managed memory and moves it with prefetch hints (memory_API.hpp:9,96-110;
// VerticesArray::move_to_device, vertices_array/gpu_api.hpp).  An MI355X pool runs with XNACK off: managed pages never migrate, they stay in host
// memory and every kernel access crosses PCIe.  This backend therefore gives every user array (VerticesArray, EdgesArray) TWO buffers, a pinned
// host mirror and a buffer in HBM, and a record of who holds the current values -- the software form of what page migration does on a CUDA box:
//
//     FRESH  --first host access-->  HOST  --kernels are about to run (or move_to_device())-->  DEVICE  --host access (or move_to_host())-->  HOST
//                                          one hipMemcpy H2D of the whole array                          one hipMemcpy D2H of the whole array
//
// * device code (the user's lambdas, which capture the arrays by value) reads and writes the HBM buffer: VerticesArray::operator[] returns
//   device_data[i] when compiled for the device;
// * host code (initialisation, the sequential checkers, reorder(), verify_results) reads and writes the mirror; every host accessor first calls
//   hip_shadow_host_access(), one load and a compare while the mirror is current;
// * GraphAbstractionsHIP calls hip_shadows_to_device() at the start of every primitive: arrays the host touched since the last one are uploaded.
//   A kernel may write any array, so after a primitive every registered array counts as newer on the device.
// move_to_device() / move_to_host() of the reference (bfs.hpp:70-74, gpu_pr.hpp:20-46, gpu_shortest_paths.hpp:220-222) are the same transitions
// made early, outside the timed region -- they stay hints: forgetting one costs a copy at the first use, never a wrong result.
// Graph containers and one-word flags (`changes[0]`) keep MemoryAPI::allocate_array = hipMallocManaged: host-resident, visible to kernels;
// the backend class traverses device copies of the adjacency.  Frontier flags / ids are device memory only (allocate_device_array): nothing but the
// backend and the containers' own mutators touches them (add_vertex: two single stores through the PCIe BAR).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

enum HipShadowOwner { HIP_SHADOW_FRESH = 0, HIP_SHADOW_HOST = 1, HIP_SHADOW_DEVICE = 2 };

struct HipShadow {
    void *host = nullptr, *device = nullptr;
    size_t bytes = 0;
    volatile int owner = HIP_SHADOW_FRESH;
};

struct HipShadowRegistry {
    std::map<void *, HipShadow *> by_host;          // every live shadowed array, keyed by its mirror (what the reference's code passes around)
    std::recursive_mutex lock;
    unsigned long long uploads = 0, downloads = 0, bytes_up = 0, bytes_down = 0;
    int not_on_device = 0;                            // arrays that are FRESH or newer on the host: hip_shadows_to_device() returns at once while this is 0
};
inline HipShadowRegistry &hip_shadow_registry();
inline void hip_shadow_print_stats()
{
    HipShadowRegistry &r = hip_shadow_registry();
    fprintf(stderr, "[vgl hip] shadowed arrays: %llu uploads (%.1f MB), %llu downloads (%.1f MB)\n", r.uploads, r.bytes_up / 1e6, r.downloads, r.bytes_down / 1e6);
}
inline HipShadowRegistry &hip_shadow_registry()
{
    static HipShadowRegistry *r = nullptr;            // (never destroyed: arrays of static duration may outlive any static registry)
    if (!r) { r = new HipShadowRegistry(); if (getenv("VGL_HIP_SHADOW_STATS")) atexit(hip_shadow_print_stats); }
    return *r;
}

#define HIP_SHADOW_RT(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw hipGetErrorString(_e); } while (0)

inline HipShadow *hip_shadow_allocate(size_t bytes)
{
    HipShadow *s = new HipShadow();
    s->bytes = bytes;
    const size_t n = bytes ? bytes : 1;
    HIP_SHADOW_RT(hipHostMalloc(&s->host, n, hipHostMallocDefault));
    // UNCACHED device memory (hipDeviceMallocUncached: not kept in the per-XCD L2s).  The reference's GPU operators are written for a GPU with ONE
    // coherent L2: gpu_shortest_paths.hpp:38-47 and gpu_shiloach_vishkin.hpp:41-50 update distances[dst] / components[dst] with plain conditional
    // stores from any thread and end when a pass stores nothing.  The MI355X has eight L2s that are not coherent with each other: two XCDs that
    // store different values to one word in a pass can each keep their own, the loop ends with every XCD content with ITS view, and memory holds
    // the larger one.  Measured with integration/tests/sssp_convergence_check.cpp on RMAT-22 (profiles/r05_binding_memkind.log): the reference's
    // operator on ordinary (coarse-grained) hipMalloc memory converged to distances that differ from seq_dijkstra for 5 of 8 sources (1 - 10 716
    // vertices), on fine-grained memory for 2 of 8, on uncached memory for none -- and in 14 - 17 passes instead of 18 - 21, at the same AVG_PERF
    // of the bfs / pr / cc apps (the Infinity Cache serves what L2 would have).  hipMalloc when the allocator refuses the flag.
    if (hipExtMallocWithFlags(&s->device, n, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        HIP_SHADOW_RT(hipMalloc(&s->device, n));
    }
    HipShadowRegistry &r = hip_shadow_registry();
    std::lock_guard<std::recursive_mutex> g(r.lock);
    r.by_host[s->host] = s;
    r.not_on_device++;
    return s;
}

inline void hip_shadow_free(HipShadow *s)
{
    if (!s) return;
    HipShadowRegistry &r = hip_shadow_registry();
    {
        std::lock_guard<std::recursive_mutex> g(r.lock);
        if (s->owner != HIP_SHADOW_DEVICE) r.not_on_device--;
        r.by_host.erase(s->host);
    }
    (void)hipDeviceSynchronize();                     // kernels that captured the array may still run
    (void)hipFree(s->device);
    (void)hipHostFree(s->host);
    delete s;
}

// the slow halves of the two transitions
inline void hip_shadow_acquire_host(HipShadow *s)
{
    HipShadowRegistry &r = hip_shadow_registry();
    std::lock_guard<std::recursive_mutex> g(r.lock);
    if (s->owner == HIP_SHADOW_HOST) return;          // another thread of the team was first
    if (s->owner == HIP_SHADOW_DEVICE && s->bytes) {
        HIP_SHADOW_RT(hipDeviceSynchronize());
        HIP_SHADOW_RT(hipMemcpy(s->host, s->device, s->bytes, hipMemcpyDeviceToHost));
        r.downloads++; r.bytes_down += s->bytes;
    }
    if (s->owner == HIP_SHADOW_DEVICE) r.not_on_device++;
    s->owner = HIP_SHADOW_HOST;
}
inline void hip_shadow_acquire_device(HipShadow *s, hipStream_t stream)
{
    HipShadowRegistry &r = hip_shadow_registry();
    std::lock_guard<std::recursive_mutex> g(r.lock);
    if (s->owner == HIP_SHADOW_DEVICE) return;
    if (s->owner == HIP_SHADOW_HOST) {
        if (s->bytes) {
            HIP_SHADOW_RT(hipMemcpyAsync(s->device, s->host, s->bytes, hipMemcpyHostToDevice, stream));
            HIP_SHADOW_RT(hipStreamSynchronize(stream));      // the mirror may be rewritten (or freed) as soon as we return
            r.uploads++; r.bytes_up += s->bytes;
        }
    }
    r.not_on_device--;
    s->owner = HIP_SHADOW_DEVICE;
}

// host code is about to read or write the mirror
inline void hip_shadow_host_access(HipShadow *s)
{
    if (__builtin_expect(s->owner != HIP_SHADOW_HOST, 0)) hip_shadow_acquire_host(s);
}
// kernels are about to run: whatever the host wrote since the last primitive goes up; afterwards every array counts as newer on the device
// (a kernel may have written any of them)
inline void hip_shadows_to_device(hipStream_t stream)
{
    HipShadowRegistry &r = hip_shadow_registry();
    if (r.not_on_device == 0) return;
    std::lock_guard<std::recursive_mutex> g(r.lock);
    for (auto &kv : r.by_host)
        if (kv.second->owner != HIP_SHADOW_DEVICE) hip_shadow_acquire_device(kv.second, stream);
}
inline HipShadow *hip_shadow_of(const void *host_pointer)
{
    HipShadowRegistry &r = hip_shadow_registry();
    std::lock_guard<std::recursive_mutex> g(r.lock);
    auto it = r.by_host.find((void *)host_pointer);
    return it == r.by_host.end() ? nullptr : it->second;
}

// VerticesArray::set_all_constant while the array lives on the device: a fill kernel instead of fetching the array back, filling the mirror and
// uploading it again at the next primitive.  Returns false (nothing done) while the host owns the array: the caller fills the mirror as before.
template <typename T>
__global__ void hip_shadow_fill_kernel(T *data, T value, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) data[i] = value;
}
template <typename T>
inline bool hip_shadow_fill(HipShadow *s, T *device_data, T value, size_t n)
{
    if (s->owner == HIP_SHADOW_HOST) return false;
    if (n) {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(hip_shadow_fill_kernel<T>, dim3(blocks), dim3(256), 0, 0, device_data, value, n);
        HIP_SHADOW_RT(hipGetLastError());
        HIP_SHADOW_RT(hipStreamSynchronize(0));
    }
    if (s->owner == HIP_SHADOW_FRESH) {
        HipShadowRegistry &r = hip_shadow_registry();
        std::lock_guard<std::recursive_mutex> g(r.lock);
        if (s->owner == HIP_SHADOW_FRESH) { s->owner = HIP_SHADOW_DEVICE; r.not_on_device--; }
    }
    return true;
}

// Graph containers that were freed or resized since the backend copied their adjacency to the device (CSRGraph::free / VectorCSRGraph::free call
// hip_container_changed): a new container at a recycled address -- same pointers from the allocator, same edge count -- must not meet the old copy.
inline std::map<const void *, unsigned long long> &hip_container_versions() { static auto *m = new std::map<const void *, unsigned long long>(); return *m; }
inline void hip_container_changed(const void *container) { hip_container_versions()[container]++; }
inline unsigned long long hip_container_version(const void *container)
{
    auto &m = hip_container_versions();
    auto it = m.find(container);
    return it == m.end() ? 0 : it->second;
}
