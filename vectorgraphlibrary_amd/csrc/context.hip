// context.hip -- context, error reporting, device memory, kernel timing hooks.
#include "vgl_hip_internal.h"
#include <unordered_set>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <algorithm>

static thread_local std::string g_last_error;

int vgl_set_error(const char *file, int line, const char *msg)
{
    char buf[512];
    const char *base = strrchr(file, '/');
    snprintf(buf, sizeof(buf), "vgl_hip: %s (%s:%d)", msg, base ? base + 1 : file, line);
    g_last_error = buf;
    return 1;
}

// the library's own stream-ordered pool of a device (vgl_pool_alloc): created once, never the device's default pool
static hipMemPool_t g_lib_pool[64];
static bool g_lib_pool_tried[64];
static std::mutex g_lib_pool_mutex;
hipMemPool_t vgl_lib_pool(int device)
{
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_lib_pool_mutex);
    if (!g_lib_pool_tried[device]) {
        g_lib_pool_tried[device] = true;
        hipMemPoolProps props;
        memset(&props, 0, sizeof(props));
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = device;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
            uint64_t keep = 48ULL << 30;
            if (const char *e = getenv("VGL_POOL_KEEP_GB")) keep = (uint64_t)std::max(0.0, atof(e)) << 30;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            g_lib_pool[device] = pool;
        } else (void)hipGetLastError();
    }
    return g_lib_pool[device];
}

// blocks that vgl_pool_alloc took from hipMalloc (see vgl_hip_internal.h)
static std::mutex g_big_mutex;
static std::unordered_set<void *> g_big_blocks;
size_t vgl_pool_block_limit()
{
    static const size_t limit = [] { const char *e = getenv("VGL_POOL_MAX_MB"); return (e ? (size_t)strtoull(e, nullptr, 10) : (size_t)64) << 20; }();
    return limit;
}
void vgl_big_block_remember(void *p) { std::lock_guard<std::mutex> lock(g_big_mutex); g_big_blocks.insert(p); }
bool vgl_big_block_forget(void *p) { std::lock_guard<std::mutex> lock(g_big_mutex); return g_big_blocks.erase(p) != 0; }

extern char **environ;
// setenv / putenv / unsetenv replace or move entries of `environ`: the pointers change, so a hash of the pointers tells whether anything was set
// since the last look without reading a single string
static uint64_t vgl_env_signature()
{
    uint64_t h = 1469598103934665603ull ^ (uint64_t)(uintptr_t)environ;
    if (environ) for (char **e = environ; *e; e++) h = (h ^ (uint64_t)(uintptr_t)*e) * 1099511628211ull;
    return h ? h : 1;
}
void vgl_ctx_refresh_env(vgl_hip_ctx *c)
{
    const uint64_t sig = vgl_env_signature();
    if (sig == c->env_signature) return;
    c->env_signature = sig;
    c->env.clear();
    if (environ)
        for (char **e = environ; *e; e++)
            if (strncmp(*e, "VGL_", 4) == 0)
                if (const char *eq = strchr(*e, '=')) c->env[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    vgl_bfs_tunables t;
    auto str = [&](const char *n) -> const char * { auto it = c->env.find(n); return it == c->env.end() ? nullptr : it->second.c_str(); };
    auto on = [&](const char *n) { const char *v = str(n); return v && v[0] == '1'; };
    if (const char *v = str("VGL_TD_EMIT_EDGES")) t.td_emit_edges = atoll(v);
    if (const char *v = str("VGL_BFS_SMALL_M")) t.small_m = atoll(v);
    if (const char *v = str("VGL_BFS_BM_EXPAND")) t.bm_expand = atoll(v);
    if (const char *v = str("VGL_SHARD_TD_EMIT_EDGES")) t.shard_td_emit_edges = atoll(v);
    if (const char *v = str("VGL_TD_FILTER_SHARE")) t.td_filter_share = atof(v);
    if (const char *v = str("VGL_TD_LATE_SHARE")) t.td_late_share = atof(v);
    if (const char *v = str("VGL_BFS_BLOCKED_SHARE")) t.blocked_share = atof(v);
    if (const char *v = str("VGL_BU_LATER_HEAVY_BLOCKS")) t.later_heavy_blocks = atoi(v);
    if (const char *v = str("VGL_SHARD_SPARSE_CAP")) t.shard_sparse_cap = std::max(0, atoi(v));
    t.no_hint = on("VGL_BFS_NO_HINT"); t.no_scan_bound = on("VGL_BFS_NO_SCAN_BOUND"); t.trace = on("VGL_BFS_TRACE"); t.bu_split = on("VGL_BU_SPLIT");
    c->bfs = t;
}
const char *vgl_env(vgl_hip_ctx *c, const char *name)
{
    vgl_ctx_refresh_env(c);
    auto it = c->env.find(name);
    return it == c->env.end() ? nullptr : it->second.c_str();
}

extern "C" {

int vgl_hip_abi_version(void) { return VGL_HIP_ABI_VERSION; }
const char *vgl_hip_last_error(void) { return g_last_error.c_str(); }

int vgl_hip_ctx_create(int device, void *stream, vgl_hip_ctx **out)
{
    if (!out) VGL_FAIL("ctx_create: null out pointer");
    int ndev = 0;
    VGL_HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) VGL_FAIL("no HIP device visible: the MI355X backend has no CPU fallback");
    if (device < 0 || device >= ndev) VGL_FAIL("ctx_create: device index out of range");
    VGL_HIP_TRY(hipSetDevice(device));
    vgl_hip_ctx *c = new vgl_hip_ctx();
    c->device = device;
    // NULL = the device's default (null) stream, which is also what torch uses unless told otherwise
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    VGL_HIP_TRY(hipMalloc((void **)&c->d_counters, sizeof(int64_t) * C_NSLOTS));
    VGL_HIP_TRY(hipHostMalloc((void **)&c->h_counters, sizeof(int64_t) * (C_NSLOTS + 8), hipHostMallocDefault));
    VGL_HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(int64_t) * C_NSLOTS, c->stream));
    VGL_HIP_TRY(hipMalloc((void **)&c->d_shards, sizeof(int64_t) * VGL_NSHARD));
    VGL_HIP_TRY(hipMemsetAsync(c->d_shards, 0, sizeof(int64_t) * VGL_NSHARD, c->stream));
    memset(c->h_counters, 0, sizeof(int64_t) * (C_NSLOTS + 8));
    vgl_ctx_refresh_env(c);
    *out = c;
    return 0;
}

int vgl_hip_ctx_trim(vgl_hip_ctx *c)
{
    if (!c) VGL_FAIL("null context");
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    if (hipMemPool_t pool = vgl_lib_pool(c->device)) VGL_HIP_TRY(hipMemPoolTrimTo(pool, 0));
    return 0;
}

int vgl_hip_ctx_destroy(vgl_hip_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (hipMemPool_t pool = vgl_lib_pool(c->device)) (void)hipMemPoolTrimTo(pool, 0);      // cached plan-build memory goes back to the device
    for (auto &kv : c->slots)
        for (auto &p : kv.second.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto e : c->event_pool) hipEventDestroy(e);
    if (c->d_partials) hipFree(c->d_partials);
    hipFree(c->d_counters);
    hipFree(c->d_shards);
    hipHostFree(c->h_counters);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int vgl_hip_ctx_sync(vgl_hip_ctx *c)
{
    if (!c) VGL_FAIL("null context");
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

void *vgl_hip_ctx_stream(vgl_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int vgl_hip_malloc(vgl_hip_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) VGL_FAIL("malloc: null argument");
    VGL_HIP_TRY(hipSetDevice(c->device));
    VGL_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
int vgl_hip_free(vgl_hip_ctx *c, void *dptr)
{
    if (!c) VGL_FAIL("free: null context");
    if (dptr) { VGL_HIP_TRY(hipStreamSynchronize(c->stream)); VGL_HIP_TRY(hipFree(dptr)); }
    return 0;
}
int vgl_hip_memcpy_h2d(vgl_hip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) VGL_FAIL("memcpy: null context");
    if (bytes == 0) return 0;
    VGL_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int vgl_hip_memcpy_d2h(vgl_hip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) VGL_FAIL("memcpy: null context");
    if (bytes == 0) return 0;
    VGL_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int vgl_hip_memset(vgl_hip_ctx *c, void *dst, int byte_value, size_t bytes)
{
    if (!c) VGL_FAIL("memset: null context");
    if (bytes == 0) return 0;
    VGL_HIP_TRY(hipMemsetAsync(dst, byte_value, bytes, c->stream));
    return 0;
}

int vgl_hip_timing_enable(vgl_hip_ctx *c, int enable)
{
    if (!c) VGL_FAIL("null context");
    c->timing = enable != 0;
    return 0;
}
// bracket only every stride-th launch that would be bracketed (1 = all): two event records cost ~5 us of stream time, three bracketed
// launches per BFS traversal are 5 % of it; a stride that is not a multiple of the launches per traversal samples every level evenly
int vgl_hip_timing_stride(vgl_hip_ctx *c, int stride)
{
    if (!c) VGL_FAIL("null context");
    c->timing_stride = stride > 1 ? stride : 1;
    c->timing_seen = 0;
    return 0;
}

int vgl_hip_timing_only(vgl_hip_ctx *c, const char *kernel_name)
{
    if (!c) VGL_FAIL("null context");
    c->timing_only = kernel_name ? kernel_name : "";
    return 0;
}
int vgl_hip_timing_reset(vgl_hip_ctx *c)
{
    if (!c) VGL_FAIL("null context");
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &kv : c->slots) {
        for (auto &p : kv.second.pending) { c->event_pool.push_back(p.first); c->event_pool.push_back(p.second); }
        kv.second.pending.clear();
        kv.second.launches = 0;
        kv.second.total_ms = 0.0;
    }
    return 0;
}
int vgl_hip_timing_get(vgl_hip_ctx *c, const char *name, int64_t *launches, double *total_ms)
{
    if (!c || !name) VGL_FAIL("null argument");
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    auto it = c->slots.find(name);
    if (it == c->slots.end()) { if (launches) *launches = 0; if (total_ms) *total_ms = 0.0; return 0; }
    vgl_timing_slot &s = it->second;
    for (auto &p : s.pending) {
        float ms = 0.f;
        VGL_HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        s.total_ms += ms;
        c->event_pool.push_back(p.first);
        c->event_pool.push_back(p.second);
    }
    s.pending.clear();
    if (launches) *launches = s.launches;
    if (total_ms) *total_ms = s.total_ms;
    return 0;
}

}  // extern "C"

static hipEvent_t vgl_get_event(vgl_hip_ctx *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

vgl_timed_launch::vgl_timed_launch(vgl_hip_ctx *c, const char *name) : ctx(c), slot(nullptr), a(nullptr), b(nullptr)
{
    if (!c->timing) return;
    if (!c->timing_only.empty() && c->timing_only != name) return;      // only the named kernel pays for its two event records
    if (c->timing_stride > 1 && (c->timing_seen++ % c->timing_stride) != 0) return;          // ... and only every stride-th of its launches
    slot = &c->slots[name];
    a = vgl_get_event(c);
    b = vgl_get_event(c);
    hipEventRecord(a, c->stream);
}
vgl_timed_launch::~vgl_timed_launch()
{
    if (!slot) return;
    hipEventRecord(b, ctx->stream);
    slot->pending.emplace_back(a, b);
    slot->launches++;
}

// counters[C_EDGES] += sum of the shards; shards are cleared
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_fold_shards(int64_t *shards, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t acc = 0;
    for (int i = threadIdx.x; i < VGL_NSHARD; i += VGL_BLOCK) { acc += shards[i]; shards[i] = 0; }
    acc = vgl_block_reduce_add(acc, s64);
    if (threadIdx.x == 0) counters[C_EDGES] += acc;
}

// One wavefront copies the counter slots into pinned host memory and then publishes a sequence number with system scope.
// The host spins on the sequence number: a few microseconds after the producing kernels finish, instead of a
// hipMemcpyAsync (blit/SDMA setup) + hipStreamSynchronize round trip per BFS level / SSSP step.
__global__ void vgl_k_publish(const int64_t *counters, volatile int64_t *host, int64_t seq)
{
    if (threadIdx.x < C_NSLOTS) host[threadIdx.x] = counters[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { host[C_NSLOTS] = seq; __threadfence_system(); }
}

int vgl_read_counters(vgl_hip_ctx *c, bool fold_shards)
{
    if (fold_shards) hipLaunchKernelGGL(vgl_k_fold_shards, dim3(1), dim3(VGL_BLOCK), 0, c->stream, c->d_shards, c->d_counters);
    const int64_t seq = ++c->publish_seq;
    hipLaunchKernelGGL(vgl_k_publish, dim3(1), dim3(64), 0, c->stream, c->d_counters, (volatile int64_t *)c->h_counters, seq);
    VGL_HIP_TRY(hipGetLastError());
    return vgl_wait_counters(c, seq);
}
int64_t vgl_next_seq(vgl_hip_ctx *c) { return ++c->publish_seq; }
int vgl_wait_counters(vgl_hip_ctx *c, int64_t seq)
{
    volatile int64_t *flag = (volatile int64_t *)c->h_counters + C_NSLOTS;
    for (long spin = 0; *flag != seq; spin++) {
        if (spin > 2000000) {                          // ~1 s: fall back to a blocking wait (surfaces launch failures too)
            VGL_HIP_TRY(hipStreamSynchronize(c->stream));
            if (*flag != seq) VGL_FAIL("counter publish did not arrive");
            break;
        }
        __builtin_ia32_pause();
    }
    return 0;
}
int vgl_zero_counters(vgl_hip_ctx *c, int first, int count)
{
    VGL_HIP_TRY(hipMemsetAsync(c->d_counters + first, 0, sizeof(int64_t) * count, c->stream));
    return 0;
}
int vgl_ensure_partials(vgl_hip_ctx *c, size_t n)
{
    if (c->partials_cap >= n) return 0;
    if (c->d_partials) { VGL_HIP_TRY(hipStreamSynchronize(c->stream)); VGL_HIP_TRY(hipFree(c->d_partials)); }
    VGL_HIP_TRY(hipMalloc((void **)&c->d_partials, sizeof(double) * n));
    c->partials_cap = n;
    return 0;
}
