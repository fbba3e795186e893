// vgl_hip.hpp -- C++ operator API of the MI355X backend: the drop-in counterpart of the reference's per-architecture
// backend class (vgl_compute_api/template/graph_abstractions_template.h:5-107, selected through
// architecture_independent_api.h:3-43).  Compile the application TU with hipcc (--offload-arch=gfx950) and link
// libvgl_hip.so; user operators are device lambdas exactly as in the reference's __USE_GPU__ flavour:
//
//     auto edge_op = [levels, cur] __VGL_SCATTER_ARGS__ { ... };
//     graph_API.scatter(graph, frontier, edge_op);
//
// Header-only part : templated HIP kernels that call the user lambdas (advance all-active / dense / sparse, compute,
//                    reduce, generate_new_frontier predicate), edge-balanced with the same LDS row-map machinery as the
//                    fused kernels of libvgl_hip.so.
// Library part     : everything that does not depend on a user type goes through the C ABI (include/vgl_hip.h):
//                    graph build, frontier compaction, scans, reductions, synthetic inputs.
//
// The data-structure classes below mirror the members the reference's algorithms and apps use
// (VGL_Graph vgl_graph.h:7-79, VGL_Frontier frontier.h:13-54, VerticesArray vertices_array.h:17-77,
//  EdgesArray edges_array.h:10-63, EdgesContainer edges_container.h:5-233).  Graph storage is plain CSR with identity
// numbering (CSR_GRAPH), so reorder() is the identity and every direction shares one vertex numbering.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <utility>
#include "../../include/vgl_hip.h"
#include "../csrc/vgl_hip_internal.h"
#include "../csrc/vgl_gnf.h"

// ------------------------------------------------------------------------------------------------------------------
// architecture macros (architecture_independent_api.h:19-43, GPU flavour)
// ------------------------------------------------------------------------------------------------------------------
#define __USE_HIP__
#define __VGL_COMPUTE_ARGS__ __device__ (int src_id, int connections_count, int vector_index)
#define __VGL_SCATTER_ARGS__ __device__ (int src_id, int dst_id, int local_edge_pos, long long int global_edge_pos, int vector_index)
#define __VGL_GATHER_ARGS__ __device__ (int src_id, int dst_id, int local_edge_pos, long long int global_edge_pos, int vector_index)
#define __VGL_ADVANCE_ARGS__ __device__ (int src_id, int dst_id, int local_edge_pos, long long int global_edge_pos, int vector_index)
#define __VGL_ADVANCE_PREPROCESS_ARGS__ __device__ (int src_id, int connections_count, int vector_index)
#define __VGL_ADVANCE_POSTPROCESS_ARGS__ __device__ (int src_id, int connections_count, int vector_index)
#define __VGL_GNF_ARGS__ __device__ (int src_id, int connections_count)->int
#define __VGL_COPY_IF_INDEXES_ARGS__ __device__ (long long idx)->int
#define __VGL_REDUCE_ANY_ARGS__ __device__ (int src_id, int connections_count, int vector_index)
#define __VGL_REDUCE_INT_ARGS__ __device__ (int src_id, int connections_count, int vector_index)->int
#define __VGL_REDUCE_FLT_ARGS__ __device__ (int src_id, int connections_count, int vector_index)->float
#define __VGL_REDUCE_DBL_ARGS__ __device__ (int src_id, int connections_count, int vector_index)->double
#define VGL_GRAPH_ABSTRACTIONS GraphAbstractionsHIP
#define VGL_FRONTIER VGL_Frontier
#include "vgl_hip_kernels.hpp"
#define VGL_SRC_ID_ADD(a, b) (vgl_src_id_add((a), (b)))
#define VGL_INC(a) (atomicAdd(&(a), 1))
#define VGL_DEC(a) (atomicSub(&(a), 1))     // architecture_independent_api.h:60
#define VGL_LAMBDA_CAP(a) a

// framework_types.h:120-160, settings.h:93
enum TraversalDirection { SCATTER = 0, GATHER = 1, ORIGINAL = 2 };
enum REDUCE_TYPE { REDUCE_SUM = 0, REDUCE_MAX = 1, REDUCE_MIN = 1, REDUCE_AVG = 3 };
enum FrontierSparsityType { ALL_ACTIVE_FRONTIER = 2, SPARSE_FRONTIER = 1, DENSE_FRONTIER = 0 };
enum DirectionType { UNDIRECTED_GRAPH = 0, DIRECTED_GRAPH = 1 };
enum DataExchangePolicy { EXCHANGE_ALL = 0, EXCHANGE_RECENTLY_CHANGED = 1, EXCHANGE_PRIVATE_DATA = 2 };   // library_data.h:5-10
// degree classes of the VectCSR advance (settings.h:57-58,99-109; apps override VECTOR_CORE_THRESHOLD_VALUE before the include): rows with
// fewer entries than the threshold form the reference's "collective" range and get the collective functor set
#ifndef VECTOR_LENGTH
#define VECTOR_LENGTH 32
#endif
#ifndef VECTOR_CORE_THRESHOLD_VALUE
#define VECTOR_CORE_THRESHOLD_VALUE VECTOR_LENGTH
#endif
#define IN_FRONTIER_FLAG 1
#define NOT_IN_FRONTIER_FLAG 0
#define MAX_WEIGHT 100

// errors are thrown as C strings, like the reference (apps/bfs/bfs.cpp:53-61)
#define VGL_HIP_CALL(expr) do { if ((expr) != 0) throw vgl_hip_last_error(); } while (0)
#define VGL_HIP_RT(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw hipGetErrorString(_e); } while (0)

// ------------------------------------------------------------------------------------------------------------------
// runtime: one context per process (VGL_RUNTIME::init_library, vgl_runtime.hpp:5-25)
// ------------------------------------------------------------------------------------------------------------------
struct VGL_RUNTIME {
    static vgl_hip_ctx *&ctx() { static vgl_hip_ctx *c = nullptr; return c; }
    static vgl_hip_comm *&comm() { static vgl_hip_comm *m = nullptr; return m; }
    static hipStream_t stream() { return (hipStream_t)vgl_hip_ctx_stream(ctx()); }
    // One process per GPU, started by any launcher that sets VGL_RANK / VGL_WORLD (the reference reads them from MPI_Init,
    // library_data/init.hpp) plus how the ranks meet: VGL_COMM_ID_FILE = a path rank 0 writes its 128-byte RCCL id to (collectives over
    // xGMI), or VGL_COMM_HOSTED = a shared-memory name (ranks may then share one GPU: tests, rehearsals).  VGL_DEVICE overrides the
    // device index (default: VGL_RANK for RCCL, 0 for the hosted transport).  Without VGL_WORLD: a world of one, no communicator.
    static void init_library(int, char **, int device = -1)
    {
        if (ctx()) return;
        const char *w = getenv("VGL_WORLD"), *r = getenv("VGL_RANK"), *idf = getenv("VGL_COMM_ID_FILE"), *hosted = getenv("VGL_COMM_HOSTED");
        const int world = w ? atoi(w) : 1, rank = r ? atoi(r) : 0;
        if (world < 1 || rank < 0 || rank >= world) throw "Error in VGL_RUNTIME::init_library : VGL_RANK / VGL_WORLD out of range";
        if (device < 0) device = getenv("VGL_DEVICE") ? atoi(getenv("VGL_DEVICE")) : (world > 1 && !hosted ? rank : 0);
        VGL_HIP_CALL(vgl_hip_ctx_create(device, nullptr, &ctx()));
        if (world > 1 || idf || hosted) {
            if (hosted) VGL_HIP_CALL(vgl_hip_comm_create_hosted(ctx(), rank, world, hosted, (size_t)4 << 20, &comm()));
            else {
                if (!idf) throw "Error in VGL_RUNTIME::init_library : VGL_WORLD > 1 needs VGL_COMM_ID_FILE or VGL_COMM_HOSTED";
                unsigned char id[VGL_HIP_COMM_ID_BYTES];
                const std::string path = idf, tmp = path + ".tmp";
                if (rank == 0) {
                    VGL_HIP_CALL(vgl_hip_comm_unique_id(id));
                    FILE *f = fopen(tmp.c_str(), "wb");
                    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) throw "Error in VGL_RUNTIME::init_library : cannot write VGL_COMM_ID_FILE";
                    fclose(f);
                    rename(tmp.c_str(), path.c_str());
                } else {
                    FILE *f = nullptr;
                    for (int tries = 0; !(f = fopen(path.c_str(), "rb")); tries++) {
                        if (tries > 12000) throw "Error in VGL_RUNTIME::init_library : rank 0's RCCL id did not appear";
                        std::this_thread::sleep_for(std::chrono::milliseconds(10));
                    }
                    const size_t got = fread(id, 1, sizeof(id), f);
                    fclose(f);
                    if (got != sizeof(id)) throw "Error in VGL_RUNTIME::init_library : short RCCL id file";
                }
                VGL_HIP_CALL(vgl_hip_comm_create(ctx(), rank, world, id, &comm()));
            }
        }
    }
    static void finalize_library()
    {
        if (comm()) { vgl_hip_comm_destroy(comm()); comm() = nullptr; }
        if (ctx()) { vgl_hip_ctx_destroy(ctx()); ctx() = nullptr; }
    }
    static void sync() { VGL_HIP_CALL(vgl_hip_ctx_sync(ctx())); }
    static int get_mpi_rank() { int r = 0; if (comm()) VGL_HIP_CALL(vgl_hip_comm_info(comm(), &r, nullptr, nullptr)); return r; }
    static int get_mpi_proc_num() { int w = 1; if (comm()) VGL_HIP_CALL(vgl_hip_comm_info(comm(), nullptr, &w, nullptr)); return w; }
};
// LibraryData (vgl_runtime/helpers/library_data/library_data.h:17-57): the rank / size accessors the reference's sources use
struct LibraryData {
    int get_mpi_rank() const { return VGL_RUNTIME::get_mpi_rank(); }
    int get_mpi_proc_num() const { return VGL_RUNTIME::get_mpi_proc_num(); }
};
static LibraryData vgl_library_data;

// MemoryAPI (memory_API.hpp:4-15): allocate_array gives HOST-VISIBLE, device-writable memory for the small flag / counter
// words the reference keeps in managed memory (e.g. `changes[0]`, gpu_shortest_paths.hpp:92-113)
struct MemoryAPI {
    // host-visible arrays alive right now: while there are any, the host may read what a primitive's operators wrote (the reference's
    // `changes[0]` pattern over managed memory), so every primitive ends with a stream synchronisation, as the reference's GPU backend does
    // (gpu/advance_csr.hpp:204).  With none alive nothing a primitive writes can be seen by the host before it calls a primitive that
    // returns a value (reduce, generate_new_frontier, frontier.size()) -- those synchronise themselves -- and advance / compute are left
    // enqueued (bfs.hpp:6-51 runs that way).  VGL_SYNC_PRIMITIVES=1 forces the synchronous behaviour (per-abstraction timers need it).
    static int &live_host_arrays() { static int n = 0; return n; }
    template <class T> static void allocate_array(T **p, size_t n) { VGL_HIP_RT(hipHostMalloc((void **)p, sizeof(T) * (n ? n : 1), hipHostMallocDefault)); live_host_arrays()++; }
    template <class T> static void free_array(T *p) { if (p) { hipHostFree(p); live_host_arrays()--; } }
    template <class T> static void allocate_device_array(T **p, size_t n) { VGL_HIP_CALL(vgl_hip_malloc(VGL_RUNTIME::ctx(), sizeof(T) * n, (void **)p)); }
    template <class T> static void free_device_array(T *p) { if (p) vgl_hip_free(VGL_RUNTIME::ctx(), p); }
};

// A few flag / counter words in DEVICE memory with a host mirror (round 4).  The reference's GPU variants keep such words in managed memory and
// write them from edge operators (`changes[0] = 1`, gpu_shortest_paths.hpp:92-113); MemoryAPI::allocate_array gives pinned HOST memory for
// that, which costs every storing wavefront a PCIe write and makes every primitive end with a stream synchronisation (live_host_arrays).
// Operators write `flags.device()[i]` instead; the host calls clear() before and fetch(i) after the primitives of a super-step.
template <int N>
struct vgl_device_words {
    int *d = nullptr;
    int h[N];
    vgl_device_words() { MemoryAPI::allocate_device_array(&d, N); clear(); }
    ~vgl_device_words() { MemoryAPI::free_device_array(d); }
    vgl_device_words(const vgl_device_words &) = delete;
    int *device() const { return d; }
    void clear() { VGL_HIP_CALL(vgl_hip_memset(VGL_RUNTIME::ctx(), d, 0, sizeof(int) * N)); }
    int fetch(int i) { VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), h, d, sizeof(int) * N)); return h[i]; }      // (synchronises)
};

// per-lane scratch "registers" of the reference's algorithm sources (vgl_compute_api/gpu/vector_register/vector_registers.h:3-70):
// VECTOR_LENGTH words in memory both sides can touch, plus the host folds over them
#define VEC_REGISTER_INT(name, value) int *reg_##name; MemoryAPI::allocate_array(&reg_##name, VECTOR_LENGTH); for (int i = 0; i < VECTOR_LENGTH; i++) reg_##name[i] = value;
#define VEC_REGISTER_FLT(name, value) float *reg_##name; MemoryAPI::allocate_array(&reg_##name, VECTOR_LENGTH); for (int i = 0; i < VECTOR_LENGTH; i++) reg_##name[i] = value;
#define VEC_REGISTER_DBL(name, value) double *reg_##name; MemoryAPI::allocate_array(&reg_##name, VECTOR_LENGTH); for (int i = 0; i < VECTOR_LENGTH; i++) reg_##name[i] = value;
template <typename _T> _T register_sum_reduce(_T *reg) { _T s = 0; for (int i = 0; i < VECTOR_LENGTH; i++) s += reg[i]; return s; }
template <typename _T> _T register_max_reduce(_T *reg) { _T m = std::numeric_limits<_T>::min(); for (int i = 0; i < VECTOR_LENGTH; i++) if (reg[i] > m) m = reg[i]; return m; }
template <typename _T> _T register_min_reduce(_T *reg) { _T m = std::numeric_limits<_T>::max(); for (int i = 0; i < VECTOR_LENGTH; i++) if (reg[i] < m) m = reg[i]; return m; }
template <typename _T> void register_free(_T *reg) { MemoryAPI::free_array(reg); }

class Timer {                                 // timer.hpp:20-56 (wall time around synchronised primitives)
    std::chrono::steady_clock::time_point t0, t1;
public:
    void start() { VGL_RUNTIME::sync(); t0 = std::chrono::steady_clock::now(); }
    void end() { VGL_RUNTIME::sync(); t1 = std::chrono::steady_clock::now(); }
    double get_time() const { return std::chrono::duration<double>(t1 - t0).count(); }
    void print_time_stats(const std::string &name) const { std::cout << name << " time: " << get_time() * 1000.0 << " ms" << std::endl; }   // timer.hpp
};
// element counts of the reference's bandwidth accounting (settings.h:140-155; apps override INT_ELEMENTS_PER_EDGE before the include)
#ifndef INT_ELEMENTS_PER_EDGE
#define INT_ELEMENTS_PER_EDGE 3.0
#endif
#ifndef COMPUTE_INT_ELEMENTS
#define COMPUTE_INT_ELEMENTS 2.0
#endif
#ifndef REDUCE_INT_ELEMENTS
#define REDUCE_INT_ELEMENTS 2.0
#endif
#ifndef GNF_INT_ELEMENTS
#define GNF_INT_ELEMENTS 1.0
#endif

// PerformanceStats (performance_stats.h:11-100, performance_stats.hpp:13-120,248-330): wall time per abstraction and the bytes the
// reference's accounting charges them (work items x INT elements x 4 B -- VGL's own model, not measured traffic), so that
// print_timers_stats() reports the same "total bandwidth / edges rate" lines a VGL user compares across backends.
// Timers per abstraction.  advance / compute are ASYNCHRONOUS unless a host-visible array is alive (MemoryAPI::live_host_arrays), so a host
// clock around them measures the enqueue and charges the kernel to whichever later call synchronises (round 3's statistics showed "reduce
// 99.6 %" for PageRank that way).  VGL_PRIMITIVE_TIMERS=1: every primitive is bracketed by a pair of HIP events on the runtime's stream instead
// and the pairs are resolved when the statistics are printed -- stream time of the primitive's own launches (~5 us of stream time per pair,
// which is why it is opt-in).  Without it the lines are labelled as host wall time.
static inline bool vgl_event_timers() { static const bool on = getenv("VGL_PRIMITIVE_TIMERS") && getenv("VGL_PRIMITIVE_TIMERS")[0] == '1'; return on; }
struct vgl_event_pairs {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    static vgl_event_pairs &get() { static vgl_event_pairs e; return e; }
};
struct vgl_stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    int ticket = -1;
    vgl_stopwatch()
    {
        if (!vgl_event_timers()) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        hipEventRecord(a, VGL_RUNTIME::stream());
        ticket = (int)vgl_event_pairs::get().pairs.size();
        vgl_event_pairs::get().pairs.emplace_back(a, b);
    }
    // host seconds, or -(ticket + 1) when the primitive is bracketed by events (resolved by PerformanceStats::resolve)
    double seconds() const
    {
        if (ticket < 0) return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        hipEventRecord(vgl_event_pairs::get().pairs[(size_t)ticket].second, VGL_RUNTIME::stream());
        return -(double)(ticket + 1);
    }
};
struct PerformanceStats {
    double inner_wall_time = 0, advance_time = 0, gather_time = 0, scatter_time = 0, compute_time = 0, reduce_time = 0, gnf_time = 0;
    size_t bytes_requested = 0, edges_visited = 0;
    std::vector<std::pair<int, int>> pending;          // (kind, ticket) of event-bracketed primitives: 0 gather, 1 scatter, 2 compute, 3 reduce, 4 gnf
    void reset_timers() { resolve(); *this = PerformanceStats(); }
    void add_time(int kind, double t)
    {
        if (t < 0) { pending.emplace_back(kind, (int)(-t) - 1); return; }
        inner_wall_time += t;
        if (kind <= 1) { advance_time += t; (kind == 0 ? gather_time : scatter_time) += t; }
        else if (kind == 2) compute_time += t;
        else if (kind == 3) reduce_time += t;
        else gnf_time += t;
    }
    void resolve()
    {
        if (pending.empty()) return;
        hipStreamSynchronize(VGL_RUNTIME::stream());
        std::vector<std::pair<int, int>> todo;
        todo.swap(pending);
        for (const auto &p : todo) {
            auto &ev = vgl_event_pairs::get().pairs[(size_t)p.second];
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) add_time(p.first, ms * 1e-3);
            hipEventDestroy(ev.first); hipEventDestroy(ev.second);
        }
    }
    void update_advance_stats(double t, size_t bytes, size_t edges, bool gather) { add_time(gather ? 0 : 1, t); bytes_requested += bytes; edges_visited += edges; }
    void update_compute_stats(double t, size_t vertices) { add_time(2, t); bytes_requested += (size_t)(vertices * COMPUTE_INT_ELEMENTS * sizeof(int)); }
    void update_reduce_stats(double t, size_t vertices) { add_time(3, t); bytes_requested += (size_t)(vertices * REDUCE_INT_ELEMENTS * sizeof(int)); }
    void update_gnf_stats(double t, size_t vertices) { add_time(4, t); bytes_requested += (size_t)(vertices * GNF_INT_ELEMENTS * sizeof(int)); }
    double get_sustained_bandwidth() { resolve(); return inner_wall_time > 0 ? bytes_requested / (inner_wall_time * 1e9) : 0.0; }    // GB/s
    double get_edges_rate() { resolve(); return inner_wall_time > 0 ? edges_visited / (inner_wall_time * 1e6) : 0.0; }               // MTEPS
    void print_timers_stats()
    {
        resolve();
        auto line = [this](const char *name, double t) {
            if (t > 0) std::cout << name << " : " << t * 1e3 << " (ms), " << (inner_wall_time > 0 ? 100.0 * t / inner_wall_time : 0.0) << " %" << std::endl;
        };
        std::cout << std::endl << (vgl_event_timers() ? "(stream time of each primitive's launches, HIP events)"
                                                      : "(host wall time per call: asynchronous primitives are charged to the next call that synchronises; VGL_PRIMITIVE_TIMERS=1 brackets them with events)")
                  << std::endl;
        line("Inner wall    ", inner_wall_time); line("Advance       ", advance_time); line("Gather        ", gather_time); line("Scatter       ", scatter_time);
        line("Compute       ", compute_time); line("Reduce        ", reduce_time); line("GNF           ", gnf_time);
        std::cout << std::endl << "total bandwidth: " << get_sustained_bandwidth() << " GB/s" << std::endl << "edges rate: " << get_edges_rate() << " MTEPS" << std::endl
                  << "edges visited: " << edges_visited << std::endl << std::endl;
    }
    double get_algorithm_performance(double t, long long edges) const { return edges / (t * 1e6); }   // MTEPS
    void print_algorithm_performance_stats(const std::string &name, double t, long long edges) const
    {
        std::cout << name << ": Wall time " << t * 1e3 << " ms, Wall (graph500) perf: " << edges / (t * 1e6) << " MTEPS" << std::endl;
    }
};
static PerformanceStats performance_stats;

// ------------------------------------------------------------------------------------------------------------------
// edges container + synthetic generators (edges_container.h, graph_generation.hpp:5-51,94-187) -- device resident
// ------------------------------------------------------------------------------------------------------------------
class EdgesContainer {
    int vertices_count = 0; long long edges_count = 0; int *src_ids = nullptr, *dst_ids = nullptr;
public:
    EdgesContainer() {}
    ~EdgesContainer() { MemoryAPI::free_device_array(src_ids); MemoryAPI::free_device_array(dst_ids); }
    EdgesContainer(const EdgesContainer &) = delete;
    void resize(int v, long long e)
    {
        MemoryAPI::free_device_array(src_ids); MemoryAPI::free_device_array(dst_ids);
        vertices_count = v; edges_count = e;
        MemoryAPI::allocate_device_array(&src_ids, (size_t)e); MemoryAPI::allocate_device_array(&dst_ids, (size_t)e);
    }
    int *get_src_ids() { return src_ids; }    // device pointers
    int *get_dst_ids() { return dst_ids; }
    int get_vertices_count() const { return vertices_count; }
    long long get_edges_count() const { return edges_count; }
    // EdgesContainer binary file (edges_container.h:58-99): int V; long long E; int type = EDGES_CONTAINER (4); int src[E]; int dst[E]
    bool save_to_binary_file(const std::string &file_name)
    {
        FILE *f = fopen(file_name.c_str(), "wb");
        if (!f) return false;
        std::vector<int> s((size_t)edges_count), d((size_t)edges_count);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), s.data(), src_ids, s.size() * sizeof(int)));
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), d.data(), dst_ids, d.size() * sizeof(int)));
        const int type = 4;
        fwrite(&vertices_count, sizeof(int), 1, f); fwrite(&edges_count, sizeof(long long), 1, f); fwrite(&type, sizeof(int), 1, f);
        fwrite(s.data(), sizeof(int), s.size(), f); fwrite(d.data(), sizeof(int), d.size(), f);
        fclose(f);
        return true;
    }
    bool load_from_binary_file(const std::string &file_name)
    {
        FILE *f = fopen(file_name.c_str(), "rb");
        if (!f) return false;
        int v = 0, type = 0; long long e = 0;
        if (fread(&v, sizeof(int), 1, f) != 1 || fread(&e, sizeof(long long), 1, f) != 1 || fread(&type, sizeof(int), 1, f) != 1) { fclose(f); return false; }
        if (type != 4) { fclose(f); throw "Error in EdgesContainer::load_from_binary_file : incorrect type of graph in file"; }
        // the kernels trust the ids: refuse counts the file cannot hold and ids outside [0, V) before anything reaches the device
        const long header = ftell(f);
        fseek(f, 0, SEEK_END);
        const long long payload = (long long)ftell(f) - header;
        fseek(f, header, SEEK_SET);
        if (v <= 0 || e < 0 || e > payload / (2 * (long long)sizeof(int))) { fclose(f); throw "Error in EdgesContainer::load_from_binary_file : corrupt header (vertex / edge counts)"; }
        std::vector<int> s((size_t)e), d((size_t)e);
        const bool ok = fread(s.data(), sizeof(int), (size_t)e, f) == (size_t)e && fread(d.data(), sizeof(int), (size_t)e, f) == (size_t)e;
        fclose(f);
        if (!ok) return false;
        for (size_t i = 0; i < (size_t)e; i++)
            if (s[i] < 0 || s[i] >= v || d[i] < 0 || d[i] >= v) throw "Error in EdgesContainer::load_from_binary_file : id out of range";
        load_from_host(v, s, d);
        return true;
    }
    void load_from_host(int v, const std::vector<int> &s, const std::vector<int> &d)
    {
        resize(v, (long long)s.size());
        VGL_HIP_CALL(vgl_hip_memcpy_h2d(VGL_RUNTIME::ctx(), src_ids, s.data(), s.size() * sizeof(int)));
        VGL_HIP_CALL(vgl_hip_memcpy_h2d(VGL_RUNTIME::ctx(), dst_ids, d.data(), d.size() * sizeof(int)));
    }
};

struct GraphGenerationAPI {
    static unsigned long long &seed() { static unsigned long long s = 1; return s; }     // deterministic (the reference seeds with time())
    static void R_MAT(EdgesContainer &ec, int v, long long e, int a, int b, int c, int d, DirectionType dir = DIRECTED_GRAPH)
    {
        int scale = 0; while ((1 << scale) < v) scale++;
        if ((1 << scale) != v) throw "R_MAT: vertices count must be a power of two";
        ec.resize(v, dir == DIRECTED_GRAPH ? e : 2 * e);
        VGL_HIP_CALL(vgl_hip_gen_rmat(VGL_RUNTIME::ctx(), scale, 0, e, seed(), a, b, c, d, 1, ec.get_src_ids(), ec.get_dst_ids()));
        if (dir != DIRECTED_GRAPH) mirror(ec, e);
    }
    static void random_uniform(EdgesContainer &ec, int v, long long e, DirectionType dir = DIRECTED_GRAPH)
    {
        int scale = 0; while ((1 << scale) < v) scale++;
        if ((1 << scale) != v) throw "random_uniform: vertices count must be a power of two";
        ec.resize(v, dir == DIRECTED_GRAPH ? e : 2 * e);
        VGL_HIP_CALL(vgl_hip_gen_uniform(VGL_RUNTIME::ctx(), scale, 0, e, seed(), ec.get_src_ids(), ec.get_dst_ids()));
        if (dir != DIRECTED_GRAPH) mirror(ec, e);
    }
private:
    static void mirror(EdgesContainer &ec, long long e)    // src[i+e] = dst[i], dst[i+e] = src[i] (graph_generation.hpp:41-49)
    {
        VGL_HIP_RT(hipMemcpyAsync(ec.get_src_ids() + e, ec.get_dst_ids(), e * sizeof(int), hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
        VGL_HIP_RT(hipMemcpyAsync(ec.get_dst_ids() + e, ec.get_src_ids(), e * sizeof(int), hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
    }
};

// ------------------------------------------------------------------------------------------------------------------
// VGL_Graph: outgoing + incoming CSR on the device (VGL_Graph::import, vgl_graph.hpp:57-68)
// ------------------------------------------------------------------------------------------------------------------
struct vgl_csr_view { const long long *rowptr; const int *adj; long long edges; };

// CSR_GRAPH: vertices keep their ids (csr/csr_graph.h).  VECTOR_CSR_GRAPH: vertices are renumbered by degree, largest first, ties by
// id (the VectCSR order of vect_csr/import.hpp:61-99) before the CSR build -- here ONE numbering by total degree shared by both
// directions (the reference sorts the outgoing and the incoming graph separately and permutes arrays in change_traversal_direction);
// the padded vector extension of the reference format is not materialised, the edge-tile kernels do not need it.
// the enumerators keep the reference's values (framework_types.h:49-57): they are written into graph files
enum GraphStorageFormat { VECTOR_CSR_GRAPH = 1, CSR_GRAPH = 3, EDGES_CONTAINER = 4 };

inline std::string get_graph_extension(GraphStorageFormat f)           // framework_types.h:87-101
{ return f == VECTOR_CSR_GRAPH ? ".vcsr" : f == CSR_GRAPH ? ".csr" : f == EDGES_CONTAINER ? ".el_container" : ".unknown"; }
inline std::string add_extension(const std::string &short_name, GraphStorageFormat f) { return short_name + get_graph_extension(f); }

// one direction of a graph file on the host (csr_graph.hpp:73-104, vect_csr_graph.hpp:141-181)
struct vgl_file_container {
    std::vector<long long> rowptr, perm; std::vector<int> adj, fwd, bwd;
    void write(FILE *f, int V, long long E, GraphStorageFormat fmt) const
    {
        const int type = (int)fmt;
        fwrite(&V, sizeof(int), 1, f); fwrite(&E, sizeof(long long), 1, f); fwrite(&type, sizeof(int), 1, f);
        fwrite(rowptr.data(), sizeof(long long), (size_t)V + 1, f); fwrite(adj.data(), sizeof(int), (size_t)E, f);
        if (fmt == VECTOR_CSR_GRAPH) { fwrite(fwd.data(), sizeof(int), (size_t)V, f); fwrite(bwd.data(), sizeof(int), (size_t)V, f); }
        fwrite(perm.data(), sizeof(long long), (size_t)E, f);
    }
    bool read(FILE *f, int V, long long E, GraphStorageFormat fmt)
    {
        int v = 0, type = 0; long long e = 0;
        if (fread(&v, sizeof(int), 1, f) != 1 || fread(&e, sizeof(long long), 1, f) != 1 || fread(&type, sizeof(int), 1, f) != 1) return false;
        if (v != V || e != E || type != (int)fmt) throw "Error in VGL_Graph::load_from_binary_file : container header does not match the file header";
        rowptr.resize((size_t)V + 1); adj.resize((size_t)E); perm.resize((size_t)E);
        bool ok = fread(rowptr.data(), sizeof(long long), (size_t)V + 1, f) == (size_t)V + 1 && fread(adj.data(), sizeof(int), (size_t)E, f) == (size_t)E;
        if (ok && fmt == VECTOR_CSR_GRAPH) {
            fwd.resize((size_t)V); bwd.resize((size_t)V);
            ok = fread(fwd.data(), sizeof(int), (size_t)V, f) == (size_t)V && fread(bwd.data(), sizeof(int), (size_t)V, f) == (size_t)V;
        }
        ok = ok && fread(perm.data(), sizeof(long long), (size_t)E, f) == (size_t)E;
        if (!ok) return false;
        // the kernels trust these arrays: refuse a file whose offsets or ids leave their ranges
        if (rowptr[0] != 0 || rowptr[(size_t)V] != E) throw "Error in VGL_Graph::load_from_binary_file : corrupt vertex pointers";
        for (int i = 0; i < V; i++) if (rowptr[(size_t)i + 1] < rowptr[(size_t)i]) throw "Error in VGL_Graph::load_from_binary_file : corrupt vertex pointers";
        for (long long i = 0; i < E; i++)
            if (adj[(size_t)i] < 0 || adj[(size_t)i] >= V || perm[(size_t)i] < 0 || perm[(size_t)i] >= E) throw "Error in VGL_Graph::load_from_binary_file : id out of range";
        for (size_t i = 0; i < fwd.size(); i++)
            if (fwd[i] < 0 || fwd[i] >= V || bwd[i] < 0 || bwd[i] >= V || bwd[(size_t)fwd[i]] != (int)i) throw "Error in VGL_Graph::load_from_binary_file : corrupt conversion arrays";
        return true;
    }
    // the VectCSR container of an edge list (vect_csr/import.hpp:61-99,257-337): vertices renumbered by THIS direction's degree
    // (largest first, stable), edges stably sorted by renumbered source; src / dst are left in that order (original ids), as the
    // reference leaves its EdgesContainer
    void build_vect_csr(int V, std::vector<int> &src, std::vector<int> &dst)
    {
        const size_t E = src.size();
        std::vector<int> deg((size_t)V, 0);
        for (size_t i = 0; i < E; i++) deg[(size_t)src[i]]++;
        bwd.resize((size_t)V); fwd.resize((size_t)V);
        for (int i = 0; i < V; i++) bwd[(size_t)i] = i;
        std::stable_sort(bwd.begin(), bwd.end(), [&](int a, int b) { return deg[(size_t)a] > deg[(size_t)b]; });
        for (int i = 0; i < V; i++) fwd[(size_t)bwd[(size_t)i]] = i;
        rowptr.assign((size_t)V + 1, 0);
        for (size_t i = 0; i < E; i++) rowptr[(size_t)fwd[(size_t)src[i]] + 1]++;
        for (int i = 0; i < V; i++) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
        std::vector<long long> cursor(rowptr.begin(), rowptr.end() - 1);
        perm.resize(E); adj.resize(E);
        for (size_t i = 0; i < E; i++) {                     // counting sort == stable sort by renumbered source
            const long long p = cursor[(size_t)fwd[(size_t)src[i]]]++;
            perm[(size_t)p] = (long long)i; adj[(size_t)p] = fwd[(size_t)dst[i]];
        }
        std::vector<int> s2(E), d2(E);
        for (size_t p = 0; p < E; p++) { s2[p] = src[(size_t)perm[p]]; d2[p] = dst[(size_t)perm[p]]; }
        src.swap(s2); dst.swap(d2);
    }
};

// A `.vcsr` graph file straight from an edge list, every stage on the device (round 5; VGL_Graph::save_to_binary_file below goes through this class's
// stored graph, which keeps ONE numbering for both directions, and rebuilds the reference's two per-direction containers on the host: minutes at
// RMAT-24).  VGL_Graph::import + save of the reference (vgl_graph.hpp:57-68,109-130, vect_csr/import.hpp:61-99,257-337): the outgoing container
// renumbers by out-degree (largest first, ties by id) and sorts the edges stably by renumbered source; the incoming container is built the same way
// from the list AS THE OUTGOING BUILD LEFT IT, transposed.  The edge list is left reordered, like the reference's EdgesContainer.
inline bool vgl_write_vect_csr_file(EdgesContainer &ec, const std::string &file_name)
{
    vgl_hip_ctx *c = VGL_RUNTIME::ctx();
    const int V = ec.get_vertices_count(); const long long E = ec.get_edges_count();
    FILE *f = fopen(file_name.c_str(), "wb");
    if (!f) return false;
    const int type = (int)VECTOR_CSR_GRAPH;
    fwrite(&V, sizeof(int), 1, f); fwrite(&E, sizeof(long long), 1, f); fwrite(&type, sizeof(int), 1, f);
    int *fwd = nullptr, *bwd = nullptr, *rs = nullptr, *rd = nullptr, *adj = nullptr, *s2 = nullptr, *d2 = nullptr;
    long long *rowptr = nullptr, *perm = nullptr;
    const size_t e1 = (size_t)std::max<long long>(E, 1);
    MemoryAPI::allocate_device_array(&fwd, (size_t)V); MemoryAPI::allocate_device_array(&bwd, (size_t)V);
    MemoryAPI::allocate_device_array(&rs, e1); MemoryAPI::allocate_device_array(&rd, e1); MemoryAPI::allocate_device_array(&adj, e1);
    MemoryAPI::allocate_device_array(&s2, e1); MemoryAPI::allocate_device_array(&d2, e1);
    MemoryAPI::allocate_device_array(&rowptr, (size_t)V + 1); MemoryAPI::allocate_device_array(&perm, e1);
    std::vector<char> host(std::max<size_t>(sizeof(long long) * e1, sizeof(long long) * ((size_t)V + 1)));
    auto put = [&](const void *device, size_t bytes) {
        if (bytes) VGL_HIP_CALL(vgl_hip_memcpy_d2h(c, host.data(), device, bytes));
        return fwrite(host.data(), 1, bytes, f) == bytes;
    };
    bool ok = true;
    int *src = ec.get_src_ids(), *dst = ec.get_dst_ids();
    for (int direction = 0; direction < 2 && ok; direction++) {
        int64_t kept = 0;
        VGL_HIP_CALL(vgl_hip_degree_order(c, V, E, src, dst, 0, fwd, bwd));                       // by the degree of THIS direction's sources
        VGL_HIP_CALL(vgl_hip_relabel_i32(c, E, fwd, src, rs));
        VGL_HIP_CALL(vgl_hip_relabel_i32(c, E, fwd, dst, rd));
        VGL_HIP_CALL(vgl_hip_coo_to_csr(c, V, E, rs, rd, 0, V, (int64_t *)rowptr, adj, (int64_t *)perm, &kept));
        if (kept != E) { fclose(f); throw "Error in vgl_write_vect_csr_file : edge list holds source ids outside [0, vertices count)"; }
        fwrite(&V, sizeof(int), 1, f); fwrite(&E, sizeof(long long), 1, f); fwrite(&type, sizeof(int), 1, f);
        ok = put(rowptr, sizeof(long long) * ((size_t)V + 1)) && put(adj, sizeof(int) * (size_t)E) && put(fwd, sizeof(int) * (size_t)V) && put(bwd, sizeof(int) * (size_t)V) &&
             put(perm, sizeof(long long) * (size_t)E);
        // the list in the order of this container (original ids), then transposed for the incoming container (EdgesContainer::transpose, vgl_graph.hpp:62)
        VGL_HIP_CALL(vgl_hip_gather_u32(c, E, (const int64_t *)perm, src, s2));
        VGL_HIP_CALL(vgl_hip_gather_u32(c, E, (const int64_t *)perm, dst, d2));
        VGL_HIP_RT(hipMemcpyAsync(src, d2, sizeof(int) * (size_t)E, hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
        VGL_HIP_RT(hipMemcpyAsync(dst, s2, sizeof(int) * (size_t)E, hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
    }
    // (after the second pass the list is back in (src, dst) orientation, in the incoming container's order -- what the reference's container holds
    // after VGL_Graph::import transposed it twice)
    for (void *q : {(void *)fwd, (void *)bwd, (void *)rs, (void *)rd, (void *)adj, (void *)s2, (void *)d2, (void *)rowptr, (void *)perm}) MemoryAPI::free_device_array((char *)q);
    return fclose(f) == 0 && ok;
}

template <class T>
__global__ void vgl_k_permute_values(int n, const int *idx, const T *in, T *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[idx[i]];
}

template <class T>
__global__ void vgl_k_fill_values(long long n, T v, T *out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = v;
}

// ParallelPrimitives::copy_if_indexes (vgl_compute_api/common/copy_if: indexes i in [0, size) with cond(i) > 0, in ascending order):
// per-workgroup counts, a scan of the (at most 4096) counts on the host, ordered writes with wave ballots
template <class Cond>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_copy_if_count(long long n, long long chunk, Cond cond, long long *counts)
{
    __shared__ long long s[VGL_BLOCK / 64];
    const long long lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    long long mine = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += VGL_BLOCK) mine += cond(i) > 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < VGL_BLOCK / 64; w++) t += s[w]; counts[blockIdx.x] = t; }
}
template <class Cond>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_copy_if_write(long long n, long long chunk, Cond cond, const long long *offsets, long long *out)
{
    __shared__ int wave_cnt[VGL_BLOCK / 64];
    const long long lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    long long base = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long start = lo; start < hi; start += VGL_BLOCK) {          // every thread of the workgroup runs every trip
        const long long i = start + threadIdx.x;
        const bool keep = i < hi && cond(i) > 0;
        const unsigned long long ballot = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(ballot);
        __syncthreads();
        long long before = 0; int total = 0;
        for (int w = 0; w < VGL_BLOCK / 64; w++) { if (w < wave) before += wave_cnt[w]; total += wave_cnt[w]; }
        if (keep) out[base + before + __popcll(ballot & ((1ULL << lane) - 1ULL))] = i;
        base += total;
        __syncthreads();
    }
}
struct ParallelPrimitives {
    template <class Cond>
    static long long copy_if_indexes(Cond cond, long long *out_indexes, long long size)
    {
        if (size <= 0) return 0;
        const int nb = (int)std::min<long long>(4096, (size + VGL_BLOCK - 1) / VGL_BLOCK);
        const long long chunk = ((size + nb - 1) / nb + VGL_BLOCK - 1) / VGL_BLOCK * VGL_BLOCK;
        long long *d_counts = nullptr;
        MemoryAPI::allocate_device_array(&d_counts, (size_t)nb);
        hipLaunchKernelGGL((vgl_k_copy_if_count<Cond>), dim3(nb), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), size, chunk, cond, d_counts);
        VGL_HIP_RT(hipGetLastError());
        std::vector<long long> h((size_t)nb);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), h.data(), d_counts, sizeof(long long) * h.size()));
        long long total = 0;
        for (auto &x : h) { const long long c = x; x = total; total += c; }
        VGL_HIP_CALL(vgl_hip_memcpy_h2d(VGL_RUNTIME::ctx(), d_counts, h.data(), sizeof(long long) * h.size()));
        if (total > 0) hipLaunchKernelGGL((vgl_k_copy_if_write<Cond>), dim3(nb), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), size, chunk, cond, d_counts, out_indexes);
        VGL_HIP_RT(hipGetLastError());
        VGL_RUNTIME::sync();
        MemoryAPI::free_device_array(d_counts);
        return total;
    }
};

class VGL_Graph {
    GraphStorageFormat format = CSR_GRAPH;
    int32_t *d_fwd = nullptr, *d_bwd = nullptr;             // VECTOR_CSR_GRAPH: original -> stored id, stored -> original id (device)
    std::vector<int> h_fwd, h_bwd;
    int vertices_count = 0; long long edges_count = 0;
    int64_t *out_rowptr = nullptr, *in_rowptr = nullptr, *out_perm = nullptr, *in_perm = nullptr;
    int32_t *out_adj = nullptr, *in_adj = nullptr;
    vgl_hip_graph *handle = nullptr;
    // layout behind the declared relax (VGL_RELAX_OVER_EDGES): built on first use, kept while the weights array stays the same
    vgl_hip_sssp_pull_plan *relax_plan = nullptr;
    const float *relax_plan_weights = nullptr;
    unsigned long long relax_plan_version = 0;
    std::vector<long long> host_out_rowptr;
public:
    explicit VGL_Graph(GraphStorageFormat f = CSR_GRAPH) : format(f) {}
    ~VGL_Graph() { release(); }
    VGL_Graph(const VGL_Graph &) = delete;
    // (version: EdgesArray::version(), bumped when the array's contents are rewritten -- the plan holds a reordered copy of the weights)
    vgl_hip_sssp_pull_plan *get_relax_plan(const float *weights, unsigned long long version = 0)
    {
        if (!relax_plan || relax_plan_weights != weights || relax_plan_version != version) {
            if (relax_plan) VGL_HIP_CALL(vgl_hip_sssp_pull_plan_destroy(VGL_RUNTIME::ctx(), relax_plan));
            relax_plan = nullptr;
            VGL_HIP_CALL(vgl_hip_sssp_pull_plan_create(VGL_RUNTIME::ctx(), handle, weights, &relax_plan));
            relax_plan_weights = weights; relax_plan_version = version;
        }
        return relax_plan;
    }
private:
    void release()
    {
        if (relax_plan) vgl_hip_sssp_pull_plan_destroy(VGL_RUNTIME::ctx(), relax_plan);
        relax_plan = nullptr; relax_plan_weights = nullptr;
        if (handle) vgl_hip_graph_destroy(VGL_RUNTIME::ctx(), handle);
        for (void *p : {(void *)out_rowptr, (void *)in_rowptr, (void *)out_perm, (void *)in_perm, (void *)out_adj, (void *)in_adj, (void *)d_fwd,
                        (void *)d_bwd})
            MemoryAPI::free_device_array((char *)p);
        handle = nullptr; out_rowptr = in_rowptr = out_perm = in_perm = nullptr; out_adj = in_adj = nullptr; d_fwd = d_bwd = nullptr;
        h_fwd.clear(); h_bwd.clear();
        mirrored = false; host_in_rowptr.clear(); host_out_adj.clear(); host_in_adj.clear();
        mpi_bounds[0].clear(); mpi_bounds[1].clear();
    }
    template <class T> static T *upload(const std::vector<T> &h)
    {
        T *d = nullptr;
        MemoryAPI::allocate_device_array(&d, std::max<size_t>(h.size(), 1));
        if (!h.empty()) VGL_HIP_CALL(vgl_hip_memcpy_h2d(VGL_RUNTIME::ctx(), d, h.data(), h.size() * sizeof(T)));
        return d;
    }
    template <class T, class D> static std::vector<T> download(const D *d, size_t n)
    {
        static_assert(sizeof(T) == sizeof(D), "download: element size");
        std::vector<T> h(n);
        if (n) VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), h.data(), d, n * sizeof(T)));
        return h;
    }
public:
    // Graph files of the reference (VGL_Graph::save_to_binary_file / load_from_binary_file, vgl_graph.hpp:109-161): header, outgoing
    // container, incoming container.  `.csr` files are this class's arrays as they are.  `.vcsr` files carry one degree numbering PER
    // DIRECTION; this class keeps one numbering for both, so saving rebuilds the two containers on the host from the stored graph
    // (the result is the file the reference writes for the same edge list, byte for byte) and loading adopts the outgoing
    // container's numbering and re-expresses the incoming container in it (rows permuted, ids mapped, adjacency order kept).
    bool save_to_binary_file(const std::string &file_name)
    {
        if (!handle) throw "Error in VGL_Graph::save_to_binary_file : the graph is empty";
        const int V = vertices_count; const long long E = edges_count;
        vgl_file_container out, in;
        out.rowptr = download<long long>(out_rowptr, (size_t)V + 1); out.adj = download<int>(out_adj, (size_t)E);
        out.perm = download<long long>(out_perm, (size_t)E);
        if (format == CSR_GRAPH) {
            in.rowptr = download<long long>(in_rowptr, (size_t)V + 1); in.adj = download<int>(in_adj, (size_t)E);
            in.perm = download<long long>(in_perm, (size_t)E);
        } else {
            std::vector<int> src((size_t)E), dst((size_t)E);             // the imported edge list: original ids, input order
            for (int r = 0; r < V; r++)
                for (long long p = out.rowptr[(size_t)r]; p < out.rowptr[(size_t)r + 1]; p++) {
                    src[(size_t)out.perm[(size_t)p]] = h_bwd[(size_t)r]; dst[(size_t)out.perm[(size_t)p]] = h_bwd[(size_t)out.adj[(size_t)p]];
                }
            out.build_vect_csr(V, src, dst);
            src.swap(dst);                                               // EdgesContainer::transpose (vgl_graph.hpp:62)
            in.build_vect_csr(V, src, dst);
        }
        FILE *f = fopen(file_name.c_str(), "wb");
        if (!f) return false;
        const int type = (int)format;
        fwrite(&V, sizeof(int), 1, f); fwrite(&E, sizeof(long long), 1, f); fwrite(&type, sizeof(int), 1, f);
        out.write(f, V, E, format); in.write(f, V, E, format);
        return fclose(f) == 0;
    }
    bool load_from_binary_file(const std::string &file_name)
    {
        FILE *f = fopen(file_name.c_str(), "rb");
        if (!f) return false;
        int V = 0, type = 0; long long E = 0;
        if (fread(&V, sizeof(int), 1, f) != 1 || fread(&E, sizeof(long long), 1, f) != 1 || fread(&type, sizeof(int), 1, f) != 1) { fclose(f); return false; }
        if ((type != (int)CSR_GRAPH && type != (int)VECTOR_CSR_GRAPH) || V <= 0 || E < 0) {
            fclose(f); throw "Error in VGL_Graph::load_from_binary_file : unsupported container type in the graph file";
        }
        const GraphStorageFormat fmt = (GraphStorageFormat)type;
        vgl_file_container out, in;
        bool ok = false;
        try { ok = out.read(f, V, E, fmt) && in.read(f, V, E, fmt); } catch (...) { fclose(f); throw; }
        fclose(f);
        if (!ok) return false;
        if (fmt != format) std::cout << "Warning! changing container type to the one of the graph file" << std::endl;    // vgl_graph.hpp:147-151
        release();
        format = fmt; vertices_count = V; edges_count = E;
        if (fmt == VECTOR_CSR_GRAPH) {
            // stored numbering := the outgoing container's; incoming row of stored vertex s = the file's row fwd_in[bwd_out[s]]
            std::vector<int> to_out((size_t)V);                           // incoming-numbering id -> stored id
            for (int x = 0; x < V; x++) to_out[(size_t)x] = out.fwd[(size_t)in.bwd[(size_t)x]];
            vgl_file_container t;
            t.rowptr.assign((size_t)V + 1, 0); t.adj.resize((size_t)E); t.perm.resize((size_t)E);
            for (int s = 0; s < V; s++) {
                const int r = in.fwd[(size_t)out.bwd[(size_t)s]];
                t.rowptr[(size_t)s + 1] = t.rowptr[(size_t)s] + (in.rowptr[(size_t)r + 1] - in.rowptr[(size_t)r]);
            }
            for (int s = 0; s < V; s++) {
                const int r = in.fwd[(size_t)out.bwd[(size_t)s]];
                long long q = t.rowptr[(size_t)s];
                for (long long p = in.rowptr[(size_t)r]; p < in.rowptr[(size_t)r + 1]; p++, q++) {
                    t.adj[(size_t)q] = to_out[(size_t)in.adj[(size_t)p]]; t.perm[(size_t)q] = in.perm[(size_t)p];
                }
            }
            in.rowptr.swap(t.rowptr); in.adj.swap(t.adj); in.perm.swap(t.perm);
            h_fwd = out.fwd; h_bwd = out.bwd;
            d_fwd = upload(h_fwd); d_bwd = upload(h_bwd);
        }
        out_rowptr = (int64_t *)upload(out.rowptr); out_adj = upload(out.adj); out_perm = (int64_t *)upload(out.perm);
        in_rowptr = (int64_t *)upload(in.rowptr); in_adj = upload(in.adj); in_perm = (int64_t *)upload(in.perm);
        VGL_HIP_CALL(vgl_hip_graph_create(VGL_RUNTIME::ctx(), V, 0, V, out_rowptr, out_adj, E, in_rowptr, in_adj, E, &handle));
        host_out_rowptr = out.rowptr;
        return true;
    }
    void import(EdgesContainer &ec)
    {
        release();
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        const int V = ec.get_vertices_count(); const long long E = ec.get_edges_count();
        vertices_count = V; edges_count = E;
        MemoryAPI::allocate_device_array(&out_rowptr, (size_t)V + 1); MemoryAPI::allocate_device_array(&in_rowptr, (size_t)V + 1);
        MemoryAPI::allocate_device_array(&out_adj, (size_t)E); MemoryAPI::allocate_device_array(&in_adj, (size_t)E);
        MemoryAPI::allocate_device_array(&out_perm, (size_t)E); MemoryAPI::allocate_device_array(&in_perm, (size_t)E);
        int64_t kept = 0;
        const int32_t *src_ids = ec.get_src_ids(), *dst_ids = ec.get_dst_ids();
        int32_t *rs = nullptr, *rd = nullptr;
        if (format == VECTOR_CSR_GRAPH) {       // renumber, then build exactly as for CSR_GRAPH (edge positions still map to INPUT edges)
            MemoryAPI::allocate_device_array(&d_fwd, (size_t)V); MemoryAPI::allocate_device_array(&d_bwd, (size_t)V);
            VGL_HIP_CALL(vgl_hip_degree_order(c, V, E, src_ids, dst_ids, 2, d_fwd, d_bwd));
            MemoryAPI::allocate_device_array(&rs, (size_t)std::max<long long>(E, 1)); MemoryAPI::allocate_device_array(&rd, (size_t)std::max<long long>(E, 1));
            VGL_HIP_CALL(vgl_hip_relabel_i32(c, E, d_fwd, src_ids, rs));
            VGL_HIP_CALL(vgl_hip_relabel_i32(c, E, d_fwd, dst_ids, rd));
            src_ids = rs; dst_ids = rd;
            h_fwd.resize((size_t)V); h_bwd.resize((size_t)V);
            VGL_HIP_CALL(vgl_hip_memcpy_d2h(c, h_fwd.data(), d_fwd, sizeof(int) * (size_t)V));
            VGL_HIP_CALL(vgl_hip_memcpy_d2h(c, h_bwd.data(), d_bwd, sizeof(int) * (size_t)V));
        }
        VGL_HIP_CALL(vgl_hip_coo_to_csr(c, V, E, src_ids, dst_ids, 0, V, out_rowptr, out_adj, out_perm, &kept));
        if (kept != E) throw "Error in VGL_Graph::import : edge list holds source ids outside [0, vertices count)";
        // the incoming container is built from the OUT-CSR-ordered list transposed (vgl_graph.hpp:61-64): src := adjacency, dst := row
        int32_t *csr_src = nullptr;
        MemoryAPI::allocate_device_array(&csr_src, (size_t)E);
        VGL_HIP_CALL(vgl_hip_gather_u32(c, E, out_perm, src_ids, csr_src));
        VGL_HIP_CALL(vgl_hip_coo_to_csr(c, V, E, out_adj, csr_src, 0, V, in_rowptr, in_adj, in_perm, &kept));
        MemoryAPI::free_device_array(csr_src);
        MemoryAPI::free_device_array(rs); MemoryAPI::free_device_array(rd);
        VGL_HIP_CALL(vgl_hip_graph_create(c, V, 0, V, out_rowptr, out_adj, E, in_rowptr, in_adj, E, &handle));
        host_out_rowptr.resize((size_t)V + 1);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(c, host_out_rowptr.data(), out_rowptr, sizeof(long long) * ((size_t)V + 1)));
    }
    int get_vertices_count() const { return vertices_count; }
    long long get_edges_count() const { return edges_count; }
    vgl_hip_graph *get_handle() const { return handle; }
    // VectorCSRGraph::get_mpi_thresholds (vect_csr/get_api.hpp:66-94): contiguous vertex ranges with ~E / ranks edges of the traversed
    // direction each (bounds on multiples of 64); every rank holds the whole graph, like the reference's MPI processes, and advances over
    // its range only.  world + 1 entries, identical on every rank.
    const std::vector<int64_t> &get_mpi_bounds(TraversalDirection d)
    {
        const int world = VGL_RUNTIME::get_mpi_proc_num(), k = d == GATHER ? 1 : 0;
        if ((int)mpi_bounds[k].size() != world + 1) {
            std::vector<int32_t> b((size_t)world + 1);
            VGL_HIP_CALL(vgl_hip_partition_rows(VGL_RUNTIME::ctx(), vertices_count, k ? in_rowptr : out_rowptr, world, b.data()));
            mpi_bounds[k].assign(b.begin(), b.end());
        }
        return mpi_bounds[k];
    }
    std::pair<int, int> get_mpi_thresholds(TraversalDirection d)
    {
        if (VGL_RUNTIME::get_mpi_proc_num() == 1) return {0, vertices_count};
        const std::vector<int64_t> &b = get_mpi_bounds(d);
        const int r = VGL_RUNTIME::get_mpi_rank();
        return {(int)b[(size_t)r], (int)b[(size_t)r + 1]};
    }
    vgl_csr_view get_direction_view(TraversalDirection d) const
    {
        return d == GATHER ? vgl_csr_view{(const long long *)in_rowptr, in_adj, edges_count}
                           : vgl_csr_view{(const long long *)out_rowptr, out_adj, edges_count};
    }
    const int64_t *get_outgoing_edges_reorder_indexes() const { return out_perm; }   // CSR position -> input edge
    const int64_t *get_incoming_edges_reorder_indexes() const { return in_perm; }    // in-CSR position -> out-CSR position
    GraphStorageFormat get_format() const { return format; }
    bool is_renumbered() const { return format == VECTOR_CSR_GRAPH; }
    const int32_t *get_forward_conversion() const { return d_fwd; }      // device, original -> stored
    const int32_t *get_backward_conversion() const { return d_bwd; }     // device, stored -> original
    // VGL_Graph::reorder(v, from, to) (vgl_graph get_api): SCATTER and GATHER share one numbering here
    int reorder(int v, TraversalDirection from, TraversalDirection to) const
    {
        if (v < 0 || v >= vertices_count) throw "Error in VGL_Graph::reorder : vertex id out of range";
        if (!is_renumbered() || (from == ORIGINAL) == (to == ORIGINAL)) return v;
        return from == ORIGINAL ? h_fwd[(size_t)v] : h_bwd[(size_t)v];
    }
    int get_outgoing_connections_count(int v) const { return (int)(host_out_rowptr[v + 1] - host_out_rowptr[v]); }   // v in stored numbering
    // host-side edge accessors of the reference (vgl_graph/get_api.hpp:14-50; its sequential checkers walk the graph through them):
    // the CSR arrays are mirrored on the host the first time one of them is called
    int get_incoming_connections_count(int v) { mirror(); return (int)(host_in_rowptr[(size_t)v + 1] - host_in_rowptr[(size_t)v]); }
    int get_outgoing_edge_dst(int v, int local_edge_pos) { mirror(); return host_out_adj[(size_t)(host_out_rowptr[(size_t)v] + local_edge_pos)]; }
    int get_incoming_edge_dst(int v, int local_edge_pos) { mirror(); return host_in_adj[(size_t)(host_in_rowptr[(size_t)v] + local_edge_pos)]; }
    // position of an edge in an EdgesArray ([outgoing E ; incoming E], csr_edges_array.hpp:67-73; vgl_graph/get_api.hpp:53-63)
    size_t get_outgoing_edges_array_index(int v, int edge_pos) const { return (size_t)(host_out_rowptr[(size_t)v] + edge_pos); }
    size_t get_incoming_edges_array_index(int v, int edge_pos) { mirror(); return (size_t)(edges_count + host_in_rowptr[(size_t)v] + edge_pos); }
private:
    std::vector<long long> host_in_rowptr; std::vector<int> host_out_adj, host_in_adj; bool mirrored = false;
    std::vector<int64_t> mpi_bounds[2];
    void mirror()
    {
        if (mirrored) return;
        host_in_rowptr = download<long long>(in_rowptr, (size_t)vertices_count + 1);
        host_out_adj = download<int>(out_adj, (size_t)edges_count); host_in_adj = download<int>(in_adj, (size_t)edges_count);
        mirrored = true;
    }
public:
    // deterministic stand-in for select_random_nz_vertex (vgl_graph get_api): k-th draw of a fixed stream; the result is an ORIGINAL
    // vertex id with outgoing edges (the same vertex whatever the storage format)
    int select_random_nz_vertex(TraversalDirection = ORIGINAL, unsigned long long draw = 0) const
    {
        unsigned long long x = 0x9E3779B97F4A7C15ULL * (draw + 1) + GraphGenerationAPI::seed();
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 29;
        int v = (int)(x % (unsigned long long)vertices_count);
        for (int i = 0; i < vertices_count; i++, v = (v + 1) % vertices_count) {
            const int s = reorder(v, ORIGINAL, SCATTER);
            if (host_out_rowptr[s + 1] > host_out_rowptr[s]) return v;
        }
        throw "select_random_nz_vertex: graph has no edges";
    }
};

// ------------------------------------------------------------------------------------------------------------------
// VerticesArray / EdgesArray: raw typed device arrays; the copy constructor is a SHALLOW alias so that lambdas can
// capture them by value (vertices_array.hpp:20-28)
// ------------------------------------------------------------------------------------------------------------------
template <typename _T>
class VerticesArray {
    _T *vertices_data = nullptr; int vertices_count = 0; bool is_copy = false; TraversalDirection direction = SCATTER;
    VGL_Graph *graph_ptr = nullptr;
public:
    VerticesArray(VGL_Graph &g, TraversalDirection d = SCATTER) : vertices_count(g.get_vertices_count()), direction(d), graph_ptr(&g)
    { MemoryAPI::allocate_device_array(&vertices_data, (size_t)vertices_count); }
    __host__ __device__ VerticesArray(const VerticesArray &o)
        : vertices_data(o.vertices_data), vertices_count(o.vertices_count), is_copy(true), direction(o.direction), graph_ptr(o.graph_ptr) {}
    __host__ __device__ ~VerticesArray()
    {
#ifndef __HIP_DEVICE_COMPILE__
        if (!is_copy && vertices_data) MemoryAPI::free_device_array(vertices_data);
#endif
    }
    __host__ __device__ inline _T &operator[](int i) const { return vertices_data[i]; }
    __host__ __device__ inline _T get(int i) const { return vertices_data[i]; }
    __host__ __device__ inline void set(int i, _T v) const { vertices_data[i] = v; }
    __host__ __device__ _T *get_ptr() const { return vertices_data; }
    int size() const { return vertices_count; }
    TraversalDirection get_direction() const { return direction; }
    void set_direction(TraversalDirection d) { direction = d; }
    // VerticesArray::reorder (vertices_array.hpp): values move between the stored numbering (SCATTER == GATHER here) and ORIGINAL
    void reorder(TraversalDirection to)
    {
        const bool from_orig = direction == ORIGINAL, to_orig = to == ORIGINAL;
        if (graph_ptr && graph_ptr->is_renumbered() && from_orig != to_orig && vertices_count > 0) {
            _T *tmp = nullptr;
            MemoryAPI::allocate_device_array(&tmp, (size_t)vertices_count);
            // to ORIGINAL: out[orig] = data[fwd[orig]];  back: out[stored] = data[bwd[stored]]
            const int *idx = to_orig ? graph_ptr->get_forward_conversion() : graph_ptr->get_backward_conversion();
            hipLaunchKernelGGL(vgl_k_permute_values<_T>, dim3((unsigned)std::min(8192, (vertices_count + 255) / 256)), dim3(256), 0, VGL_RUNTIME::stream(),
                               vertices_count, idx, (const _T *)vertices_data, tmp);
            VGL_HIP_RT(hipGetLastError());
            VGL_HIP_RT(hipMemcpyAsync(vertices_data, tmp, sizeof(_T) * (size_t)vertices_count, hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
            VGL_RUNTIME::sync();
            MemoryAPI::free_device_array(tmp);
        }
        direction = to;
    }
    // component labels (vertex ids of the stored numbering, smallest member per component) -> smallest ORIGINAL id per component
    void reorder_labels_to_original()
    {
        static_assert(sizeof(_T) == 4, "labels are 32-bit vertex ids");
        if (graph_ptr && graph_ptr->is_renumbered() && direction != ORIGINAL && vertices_count > 0) {
            int32_t *scratch = nullptr, *out = nullptr;
            MemoryAPI::allocate_device_array(&scratch, (size_t)vertices_count); MemoryAPI::allocate_device_array(&out, (size_t)vertices_count);
            VGL_HIP_CALL(vgl_hip_cc_labels_to_original(VGL_RUNTIME::ctx(), vertices_count, (const int32_t *)vertices_data, graph_ptr->get_forward_conversion(),
                                                       graph_ptr->get_backward_conversion(), scratch, out));
            VGL_HIP_RT(hipMemcpyAsync(vertices_data, out, sizeof(_T) * (size_t)vertices_count, hipMemcpyDeviceToDevice, VGL_RUNTIME::stream()));
            VGL_RUNTIME::sync();
            MemoryAPI::free_device_array(scratch); MemoryAPI::free_device_array(out);
        }
        direction = ORIGINAL;
    }
    void set_all_constant(_T v)
    {
        std::vector<_T> h((size_t)vertices_count, v);
        VGL_HIP_CALL(vgl_hip_memcpy_h2d(VGL_RUNTIME::ctx(), vertices_data, h.data(), sizeof(_T) * h.size()));
    }
    std::vector<_T> to_host() const
    {
        std::vector<_T> h((size_t)vertices_count);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), h.data(), vertices_data, sizeof(_T) * h.size()));
        return h;
    }
};

template <typename _T>
class EdgesArray {                     // layout [outgoing E ; incoming E] (csr_edges_array.hpp:67-73)
    _T *edges_data = nullptr; long long edges_count = 0; bool is_copy = false; VGL_Graph *graph_ptr = nullptr;
    unsigned long long version_ = 0;       // bumped by every host-side writer: caches built from the values (SSSP plans) key on it
public:
    unsigned long long version() const { return version_; }
    EdgesArray(VGL_Graph &g) : edges_count(g.get_edges_count()), graph_ptr(&g)
    { MemoryAPI::allocate_device_array(&edges_data, (size_t)(2 * edges_count)); }
    __host__ __device__ EdgesArray(const EdgesArray &o) : edges_data(o.edges_data), edges_count(o.edges_count), is_copy(true), graph_ptr(o.graph_ptr), version_(o.version_) {}
    __host__ __device__ ~EdgesArray()
    {
#ifndef __HIP_DEVICE_COMPILE__
        if (!is_copy && edges_data) MemoryAPI::free_device_array(edges_data);
#endif
    }
    __host__ __device__ inline _T &operator[](long long i) const { return edges_data[i]; }
    __host__ __device__ inline _T get(long long i) const { return edges_data[i]; }
    __host__ __device__ _T *get_ptr() const { return edges_data; }
    // random f32 in [0, max) per INPUT edge, carried to both CSR orders (csr_edges_array.hpp:31-40); deterministic stream
    void set_all_random(_T max_rand)
    {
        static_assert(sizeof(_T) == 4, "4-byte edge properties");
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        float *w_in = nullptr;
        MemoryAPI::allocate_device_array(&w_in, (size_t)edges_count);
        VGL_HIP_CALL(vgl_hip_gen_weights(c, 0, edges_count, GraphGenerationAPI::seed(), w_in));   // uniform [0,100)
        (void)max_rand;
        VGL_HIP_CALL(vgl_hip_gather_u32(c, edges_count, graph_ptr->get_outgoing_edges_reorder_indexes(), w_in, edges_data));
        VGL_HIP_CALL(vgl_hip_gather_u32(c, edges_count, graph_ptr->get_incoming_edges_reorder_indexes(), edges_data, edges_data + edges_count));
        VGL_RUNTIME::sync();
        MemoryAPI::free_device_array(w_in);
        version_++;
    }
    void set_all_constant(_T v)            // both halves (csr_edges_array.hpp)
    {
        hipLaunchKernelGGL(vgl_k_fill_values<_T>, dim3(1024), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), 2 * edges_count, v, edges_data);
        VGL_HIP_RT(hipGetLastError());
        VGL_RUNTIME::sync();
        version_++;
    }
    void finalize_advance() {}             // NEC-only merge of per-core copies (tc.hpp:77-78): nothing to do here
    template <class Merge> void finalize_advance(Merge &&) {}
    std::vector<_T> outgoing_to_host() const
    {
        std::vector<_T> h((size_t)edges_count);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), h.data(), edges_data, sizeof(_T) * h.size()));
        return h;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// VGL_Frontier (frontier.h:13-54 / base_frontier.h:5-62)
// ------------------------------------------------------------------------------------------------------------------
class VGL_Frontier {
    vgl_hip_frontier *handle = nullptr; VGL_Graph *graph_ptr; TraversalDirection direction;
public:
    VGL_Frontier(VGL_Graph &g, TraversalDirection d = SCATTER) : graph_ptr(&g), direction(d)
    { VGL_HIP_CALL(vgl_hip_frontier_create(VGL_RUNTIME::ctx(), g.get_handle(), &handle)); }
    ~VGL_Frontier() { if (handle) vgl_hip_frontier_destroy(VGL_RUNTIME::ctx(), handle); }
    VGL_Frontier(const VGL_Frontier &) = delete;
    vgl_hip_frontier *get_handle() const { return handle; }
    void set_all_active() { VGL_HIP_CALL(vgl_hip_frontier_set_all_active(VGL_RUNTIME::ctx(), handle)); }
    void clear() { VGL_HIP_CALL(vgl_hip_frontier_clear(VGL_RUNTIME::ctx(), handle)); }
    void add_vertex(int v) { VGL_HIP_CALL(vgl_hip_frontier_add_vertex(VGL_RUNTIME::ctx(), handle, v)); }
    int size() const { int32_t s; VGL_HIP_CALL(vgl_hip_frontier_info(VGL_RUNTIME::ctx(), handle, &s, nullptr, nullptr)); return s; }
    long long get_neighbours_count() const { int64_t n; VGL_HIP_CALL(vgl_hip_frontier_info(VGL_RUNTIME::ctx(), handle, nullptr, &n, nullptr)); return n; }
    FrontierSparsityType get_sparsity_type() const { int t; VGL_HIP_CALL(vgl_hip_frontier_info(VGL_RUNTIME::ctx(), handle, nullptr, nullptr, &t)); return (FrontierSparsityType)t; }
    int *get_ids() const { return vgl_hip_frontier_ids(handle); }        // device
    int *get_flags() const { return vgl_hip_frontier_flags(handle); }    // device
    TraversalDirection get_direction() const { return direction; }
    void set_direction(TraversalDirection d) { direction = d; }
    void reorder(TraversalDirection) {}
};

struct vgl_empty_vertex_op { __device__ void operator()(int, int, int) const {} };
struct vgl_empty_edge_op { __device__ void operator()(int, int, int, long long, int) const {} };
static const vgl_empty_vertex_op EMPTY_VERTEX_OP;
static const vgl_empty_edge_op EMPTY_EDGE_OP;

// ------------------------------------------------------------------------------------------------------------------
// GraphAbstractionsHIP: public member list of graph_abstractions_template.h:44-104
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// DECLARED operators -- an extension, not part of the reference API.  A lambda is opaque: the backend must run it per edge in CSR order with
// whatever gathers and atomics it contains.  When the caller STATES the algebra of an all-active advance instead, the backend may pick the
// layout: VGL_MIN_LABEL_OVER_EDGES(labels) says "labels[dst] = min(labels[dst], labels[src]) for every edge" (the Shiloach-Vishkin hook,
// shiloach_vishkin.hpp:37-50) and scatter() runs it as the library's blocked pass (vgl_blocked.h: both label windows in LDS, 4 - 12 B per
// edge streamed, no atomic per edge) -- the same fixed point as the lambda form, reached in passes that see the labels of the pass start.
// ------------------------------------------------------------------------------------------------------------------
struct vgl_declared_min_label { int *labels; };
inline vgl_declared_min_label VGL_MIN_LABEL_OVER_EDGES(VerticesArray<int> &labels) { return vgl_declared_min_label{labels.get_ptr()}; }
// VGL_RELAX_OVER_EDGES(distances, weights): "distances[dst] = min(distances[dst], distances[src] + weights[edge]) for every edge" (the Bellman-Ford
// relax, shortest_paths.hpp:123-133).  The graph object keeps a blocked layout of (adjacency, weights) for the weight array last used (built on first use).
struct vgl_declared_relax { float *distances; const float *weights; unsigned long long weights_version; };
inline vgl_declared_relax VGL_RELAX_OVER_EDGES(VerticesArray<float> &distances, EdgesArray<float> &weights)
{ return vgl_declared_relax{distances.get_ptr(), weights.get_ptr(), weights.version()}; }
// VGL_SUM_OVER_EDGES(sums, values[, bound]): "sums[src] = sum of values[dst] over every edge src -> dst with dst != src" (the pull of PageRank,
// pr.hpp:109-123, whose edge operator is `page_ranks[src] += old_rank[dst] * reversed_degree[dst]` outside self loops).  Runs as the blocked pass:
// values from LDS windows, exact fixed-point accumulation (the same bits for any schedule; <= 1e-6 of the f32 chain while rows are short).
// values must be non-negative, every per-vertex sum at most `bound`.  Pre / post vertex operators are honoured around it.
struct vgl_declared_sum { float *sums; const float *values; float bound; };
inline vgl_declared_sum VGL_SUM_OVER_EDGES(VerticesArray<float> &sums, VerticesArray<float> &values, float bound = 1.0f)
{ return vgl_declared_sum{sums.get_ptr(), values.get_ptr(), bound}; }

class GraphAbstractionsHIP {
    VGL_Graph *processed_graph_ptr; TraversalDirection current_traversal_direction;
    double *reduce_partials = nullptr;          // one per workgroup of a reduce (+ the folded maximum)
    bool sequential_rows = false;               // enable_sequential_rows()

    static int active_count(VGL_Graph &g, VGL_Frontier &f) { return f.get_sparsity_type() == ALL_ACTIVE_FRONTIER ? g.get_vertices_count() : f.size(); }
    static unsigned grid_for(long long n) { long long b = (n + VGL_BLOCK - 1) / VGL_BLOCK; return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }
    template <class T> static constexpr bool is_empty_vertex_op() { return std::is_same<typename std::decay<T>::type, vgl_empty_vertex_op>::value; }

    // see MemoryAPI::live_host_arrays
    static bool sync_after_primitive()
    {
        static const bool forced = getenv("VGL_SYNC_PRIMITIVES") && getenv("VGL_SYNC_PRIMITIVES")[0] == '1';
        return forced || MemoryAPI::live_host_arrays() > 0;
    }
    void set_correct_direction() {}
    template <typename _T, typename... Types>
    void set_correct_direction(_T &first, Types &...rest) { first.reorder(current_traversal_direction); first.set_direction(current_traversal_direction); set_correct_direction(rest...); }

    template <class Op>
    void vertex_pass(VGL_Graph &g, VGL_Frontier &f, TraversalDirection dir, Op &&op, int row_lo = 0, int row_hi = 0x7fffffff)
    {
        const vgl_csr_view v = g.get_direction_view(dir);
        const FrontierSparsityType t = f.get_sparsity_type();
        hipStream_t st = VGL_RUNTIME::stream();
        const int V = g.get_vertices_count();
        if (t == ALL_ACTIVE_FRONTIER) hipLaunchKernelGGL((vgl_k_vertex_op<0, typename std::decay<Op>::type>), dim3(grid_for(V)), dim3(VGL_BLOCK), 0, st, V, v.rowptr, f.get_flags(), f.get_ids(), row_lo, row_hi, op);
        else if (t == DENSE_FRONTIER) hipLaunchKernelGGL((vgl_k_vertex_op<1, typename std::decay<Op>::type>), dim3(grid_for(V)), dim3(VGL_BLOCK), 0, st, V, v.rowptr, f.get_flags(), f.get_ids(), row_lo, row_hi, op);
        else if (f.size() > 0) hipLaunchKernelGGL((vgl_k_vertex_op<2, typename std::decay<Op>::type>), dim3(grid_for(f.size())), dim3(VGL_BLOCK), 0, st, f.size(), v.rowptr, f.get_flags(), f.get_ids(), row_lo, row_hi, op);
        VGL_HIP_RT(hipGetLastError());
    }

    // advance_worker: pre (per vertex) -> edge_op over every edge of the active vertices -> post (per vertex).
    // Kernel boundaries give the per-vertex ordering the reference guarantees (advance_worker.hpp:79-99).
    template <class EdgeOp, class PreOp, class PostOp>
    void advance_worker(VGL_Graph &g, VGL_Frontier &f, TraversalDirection dir, EdgeOp &&edge_op, PreOp &&pre_op, PostOp &&post_op)
    {
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        hipStream_t st = VGL_RUNTIME::stream();
        const vgl_stopwatch watch;
        long long work = 0;                                 // edges the frontier touches (advance_worker.hpp:140-149)
        const vgl_csr_view v = g.get_direction_view(dir);
        const long long process_shift = (dir == GATHER) ? g.get_edges_count() : 0;     // compute_process_shift (graph_abstractions.hpp:19-28)
        // several ranks (inner_mpi_processing, common/advance.hpp:28-31, nec/advance_worker.hpp:239-251): this rank's vertex range only
        const std::pair<int, int> range = g.get_mpi_thresholds(dir);
        const int row_lo = range.first, row_hi = range.second;
        const FrontierSparsityType t = f.get_sparsity_type();
        using E = typename std::decay<EdgeOp>::type;
        if (sequential_rows) {
            using P = typename std::decay<PreOp>::type; using Q = typename std::decay<PostOp>::type;
            const int V = g.get_vertices_count();
            work = t == ALL_ACTIVE_FRONTIER ? v.edges : f.get_neighbours_count();
            if (t == ALL_ACTIVE_FRONTIER) hipLaunchKernelGGL((vgl_k_advance_rows<0, E, P, Q>), dim3(grid_for(V)), dim3(VGL_BLOCK), 0, st, V, v.rowptr, v.adj, f.get_flags(), f.get_ids(), process_shift, row_lo, row_hi, edge_op, pre_op, post_op);
            else if (t == DENSE_FRONTIER) hipLaunchKernelGGL((vgl_k_advance_rows<1, E, P, Q>), dim3(grid_for(V)), dim3(VGL_BLOCK), 0, st, V, v.rowptr, v.adj, f.get_flags(), f.get_ids(), process_shift, row_lo, row_hi, edge_op, pre_op, post_op);
            else if (f.size() > 0) hipLaunchKernelGGL((vgl_k_advance_rows<2, E, P, Q>), dim3(grid_for(f.size())), dim3(VGL_BLOCK), 0, st, f.size(), v.rowptr, v.adj, f.get_flags(), f.get_ids(), process_shift, row_lo, row_hi, edge_op, pre_op, post_op);
            VGL_HIP_RT(hipGetLastError());
            if (sync_after_primitive()) VGL_RUNTIME::sync();
            performance_stats.update_advance_stats(watch.seconds(), (size_t)(work * INT_ELEMENTS_PER_EDGE * sizeof(int)), (size_t)work, dir == GATHER);
            return;
        }
        if (!is_empty_vertex_op<PreOp>()) vertex_pass(g, f, dir, pre_op, row_lo, row_hi);
        if (t == SPARSE_FRONTIER) {
            const int64_t *offs; const int32_t *tile_first; int64_t M;
            VGL_HIP_CALL(vgl_hip_frontier_advance_plan(c, g.get_handle(), f.get_handle(), dir == GATHER, &offs, &tile_first, &M));
            work = M;
            if (M > 0) {
                const unsigned nt = (unsigned)((M + VGL_TILE - 1) / VGL_TILE);
                hipLaunchKernelGGL((vgl_k_advance_sparse<E>), dim3(nt), dim3(VGL_ADV_THREADS), 0, st, f.get_ids(), offs, tile_first, f.size(), (long long)M,
                                   v.rowptr, v.adj, process_shift, row_lo, row_hi, edge_op);
            }
        } else if (v.edges > 0) {
            work = t == ALL_ACTIVE_FRONTIER ? v.edges : f.get_neighbours_count();
            const int32_t *tile_row; int64_t ntiles;
            VGL_HIP_CALL(vgl_hip_graph_tile_rows(g.get_handle(), dir == GATHER, &tile_row, &ntiles));
            if (t == DENSE_FRONTIER)
                hipLaunchKernelGGL((vgl_k_advance_static<true, E>), dim3((unsigned)ntiles), dim3(VGL_BLOCK), 0, st, v.rowptr, v.adj, tile_row, v.edges, process_shift, f.get_flags(), row_lo, row_hi, edge_op);
            else
                hipLaunchKernelGGL((vgl_k_advance_static<false, E>), dim3((unsigned)ntiles), dim3(VGL_BLOCK), 0, st, v.rowptr, v.adj, tile_row, v.edges, process_shift, f.get_flags(), row_lo, row_hi, edge_op);
        }
        VGL_HIP_RT(hipGetLastError());
        if (!is_empty_vertex_op<PostOp>()) vertex_pass(g, f, dir, post_op, row_lo, row_hi);
        if (sync_after_primitive()) VGL_RUNTIME::sync();     // synchronous like the reference GPU backend (advance_csr.hpp:204) whenever the host could see a result
        performance_stats.update_advance_stats(watch.seconds(), (size_t)(work * INT_ELEMENTS_PER_EDGE * sizeof(int)), (size_t)work, dir == GATHER);
    }

    // the six-functor form (common/advance.hpp:6-115).  CSR_GRAPH: the reference's CSR worker never calls the collective set
    // (advance_worker.hpp:62-149) -- neither does this one.  VECTOR_CSR_GRAPH: rows shorter than VECTOR_CORE_THRESHOLD_VALUE get the
    // collective set, with two documented differences from advance_{all_active,dense}.hpp: global_edge_pos is the CSR position
    // (process_shift + position, as in the reference's own sparse collective kernel, advance_sparse.hpp:149) because no padded
    // vector-extension copy of the edge arrays exists here, and pre / post run for active rows only.
    template <class EdgeOp, class PreOp, class PostOp, class CEdgeOp, class CPreOp, class CPostOp>
    void advance_six(VGL_Graph &g, VGL_Frontier &f, TraversalDirection dir, EdgeOp &&edge_op, PreOp &&pre, PostOp &&post, CEdgeOp &&c_edge_op,
                     CPreOp &&c_pre, CPostOp &&c_post)
    {
        using E = typename std::decay<EdgeOp>::type; using CE = typename std::decay<CEdgeOp>::type;
        using P = typename std::decay<PreOp>::type; using CP = typename std::decay<CPreOp>::type;
        using Q = typename std::decay<PostOp>::type; using CQ = typename std::decay<CPostOp>::type;
        constexpr bool same_set = std::is_same<E, CE>::value && std::is_same<P, CP>::value && std::is_same<Q, CQ>::value;
        if (g.get_format() != VECTOR_CSR_GRAPH || same_set) { advance_worker(g, f, dir, edge_op, pre, post); return; }
        const int thr = VECTOR_CORE_THRESHOLD_VALUE;
        const vgl_csr_view v = g.get_direction_view(dir);
        const vgl_split_edge_op<E, CE> e2{edge_op, c_edge_op, v.rowptr, thr};
        if (is_empty_vertex_op<PreOp>() && is_empty_vertex_op<CPreOp>() && is_empty_vertex_op<PostOp>() && is_empty_vertex_op<CPostOp>())
            advance_worker(g, f, dir, e2, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP);
        else
            advance_worker(g, f, dir, e2, vgl_split_vertex_op<P, CP>{pre, c_pre, thr}, vgl_split_vertex_op<Q, CQ>{post, c_post, thr});
    }
    std::vector<void *> user_data_containers;

public:
    // GraphAbstractions::attach_data (graph_abstractions.hpp:130-133): registers a user array with the abstraction object
    template <typename _T> void attach_data(VerticesArray<_T> &array) { user_data_containers.push_back((void *)&array); }

    GraphAbstractionsHIP(VGL_Graph &g, TraversalDirection initial = SCATTER) : processed_graph_ptr(&g), current_traversal_direction(initial)
    { MemoryAPI::allocate_device_array(&reduce_partials, 1024 + 8); }
    ~GraphAbstractionsHIP()
    {
        MemoryAPI::free_device_array(reduce_partials);
        MemoryAPI::free_device_array((char *)exchange_scratch); MemoryAPI::free_array(exchange_heads);
    }

    // change_traversal_direction (graph_abstractions.hpp:87-125): tags and permutes every passed container; with identity
    // numbering only the tag changes
    template <typename... Types>
    void change_traversal_direction(TraversalDirection d, Types &...args) { current_traversal_direction = d; set_correct_direction(args...); }

    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation,
              typename CollectiveEdgeOperation, typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void scatter(VGL_Graph &g, VGL_Frontier &f, EdgeOperation &&edge_op, VertexPreprocessOperation &&pre, VertexPostprocessOperation &&post,
                 CollectiveEdgeOperation &&c_edge_op, CollectiveVertexPreprocessOperation &&c_pre, CollectiveVertexPostprocessOperation &&c_post)
    {
        if (current_traversal_direction != SCATTER) throw "VGL ERROR: incorrect traversal direction in scatter";   // common/advance.hpp:19-26
        advance_six(g, f, SCATTER, edge_op, pre, post, c_edge_op, c_pre, c_post);
    }
    template <typename EdgeOperation>
    void scatter(VGL_Graph &g, VGL_Frontier &f, EdgeOperation &&edge_op)
    {
        if (current_traversal_direction != SCATTER) throw "VGL ERROR: incorrect traversal direction in scatter";
        advance_worker(g, f, SCATTER, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP);
    }
    // a declared operator over an ALL_ACTIVE frontier (see VGL_MIN_LABEL_OVER_EDGES): returns whether any label changed (host value: the call synchronises)
    bool scatter(VGL_Graph &g, VGL_Frontier &f, vgl_declared_relax op)
    {
        if (current_traversal_direction != SCATTER) throw "VGL ERROR: incorrect traversal direction in scatter";
        if (f.get_sparsity_type() != ALL_ACTIVE_FRONTIER) throw "VGL ERROR: a declared operator needs an all-active frontier";
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        vgl_hip_sssp_pull_plan *plan = g.get_relax_plan(op.weights, op.weights_version);       // (kept by the graph: the next call with the same weights pays nothing)
        const vgl_stopwatch watch;
        int changed = 0;
        VGL_HIP_CALL(vgl_hip_sssp_pull_pass(c, g.get_handle(), plan, op.distances, &changed));
        const long long work = g.get_direction_view(SCATTER).edges;
        performance_stats.update_advance_stats(watch.seconds(), (size_t)(work * INT_ELEMENTS_PER_EDGE * sizeof(int)), (size_t)work, false);
        return changed != 0;
    }
    template <class PreOp, class PostOp>
    void scatter(VGL_Graph &g, VGL_Frontier &f, vgl_declared_sum op, PreOp &&pre_op, PostOp &&post_op)
    {
        if (current_traversal_direction != SCATTER) throw "VGL ERROR: incorrect traversal direction in scatter";
        if (f.get_sparsity_type() != ALL_ACTIVE_FRONTIER) throw "VGL ERROR: a declared operator needs an all-active frontier";
        const vgl_stopwatch watch;
        if (!is_empty_vertex_op<PreOp>()) vertex_pass(g, f, SCATTER, pre_op);
        VGL_HIP_CALL(vgl_hip_sum_over_edges_f32(VGL_RUNTIME::ctx(), g.get_handle(), op.values, op.bound, op.sums));
        if (!is_empty_vertex_op<PostOp>()) vertex_pass(g, f, SCATTER, post_op);
        if (sync_after_primitive()) VGL_RUNTIME::sync();
        const long long work = g.get_direction_view(SCATTER).edges;
        performance_stats.update_advance_stats(watch.seconds(), (size_t)(work * INT_ELEMENTS_PER_EDGE * sizeof(int)), (size_t)work, false);
    }
    void scatter(VGL_Graph &g, VGL_Frontier &f, vgl_declared_sum op) { scatter(g, f, op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP); }
    // prepare(declared operator): builds NOW the blocked layout the operator's first scatter would otherwise build inside the caller's timer (the
    // counterpart of what the reference keeps out of its timers: import, move_to_device; the library's own legs report their plans apart too).
    // Idempotent; returns the seconds it took.
    double prepare(VGL_Graph &g, vgl_declared_sum)
    {
        const auto t0 = std::chrono::steady_clock::now();
        VGL_HIP_CALL(vgl_hip_pr_prepare(VGL_RUNTIME::ctx(), g.get_handle(), VGL_HIP_PR_BLOCKED, nullptr));
        VGL_RUNTIME::sync();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    double prepare(VGL_Graph &g, vgl_declared_min_label)
    {
        const auto t0 = std::chrono::steady_clock::now();
        VGL_HIP_CALL(vgl_hip_cc_prepare(VGL_RUNTIME::ctx(), g.get_handle()));
        VGL_RUNTIME::sync();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    double prepare(VGL_Graph &g, vgl_declared_relax op)
    {
        const auto t0 = std::chrono::steady_clock::now();
        g.get_relax_plan(op.weights, op.weights_version);
        VGL_RUNTIME::sync();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    bool scatter(VGL_Graph &g, VGL_Frontier &f, vgl_declared_min_label op)
    {
        if (current_traversal_direction != SCATTER) throw "VGL ERROR: incorrect traversal direction in scatter";
        if (f.get_sparsity_type() != ALL_ACTIVE_FRONTIER) throw "VGL ERROR: a declared operator needs an all-active frontier";
        const vgl_stopwatch watch;
        int changed = 0;
        VGL_HIP_CALL(vgl_hip_cc_hook_owned(VGL_RUNTIME::ctx(), g.get_handle(), op.labels, &changed));
        const long long work = g.get_direction_view(SCATTER).edges;
        performance_stats.update_advance_stats(watch.seconds(), (size_t)(work * INT_ELEMENTS_PER_EDGE * sizeof(int)), (size_t)work, false);
        return changed != 0;
    }
    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation,
              typename CollectiveEdgeOperation, typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void gather(VGL_Graph &g, VGL_Frontier &f, EdgeOperation &&edge_op, VertexPreprocessOperation &&pre, VertexPostprocessOperation &&post,
                CollectiveEdgeOperation &&c_edge_op, CollectiveVertexPreprocessOperation &&c_pre, CollectiveVertexPostprocessOperation &&c_post)
    {
        if (current_traversal_direction != GATHER) throw "VGL ERROR: incorrect traversal direction in gather";
        advance_six(g, f, GATHER, edge_op, pre, post, c_edge_op, c_pre, c_post);
    }
    template <typename EdgeOperation>
    void gather(VGL_Graph &g, VGL_Frontier &f, EdgeOperation &&edge_op)
    {
        if (current_traversal_direction != GATHER) throw "VGL ERROR: incorrect traversal direction in gather";
        advance_worker(g, f, GATHER, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP);
    }

    template <typename ComputeOperation>
    void compute(VGL_Graph &g, VGL_Frontier &f, ComputeOperation &&compute_op)
    {
        const vgl_stopwatch watch;
        vertex_pass(g, f, current_traversal_direction, compute_op);
        if (sync_after_primitive()) VGL_RUNTIME::sync();
        performance_stats.update_compute_stats(watch.seconds(), (size_t)active_count(g, f));
    }

    template <typename _T, typename ReduceOperation>
    _T reduce(VGL_Graph &g, VGL_Frontier &f, ReduceOperation &&reduce_op, REDUCE_TYPE type)
    {
        if (type != REDUCE_SUM && type != REDUCE_MAX) throw "Error in GraphAbstractionsHIP::reduce : unsupported reduce type";   // reduce.hpp:144-150
        const vgl_stopwatch watch;
        const vgl_csr_view v = g.get_direction_view(current_traversal_direction);
        const FrontierSparsityType t = f.get_sparsity_type();
        hipStream_t st = VGL_RUNTIME::stream();
        using R = typename std::decay<ReduceOperation>::type;
        const int V = g.get_vertices_count();
        const int n = t == SPARSE_FRONTIER ? f.size() : V;
        if (n <= 0) return (_T)0;
        const int nb = (int)std::min<long long>(1024, ((long long)n + VGL_BLOCK - 1) / VGL_BLOCK);
        const bool mx = type == REDUCE_MAX;
#define VGL_REDUCE_LAUNCH(MODE)                                                                                                                          \
        do {                                                                                                                                             \
            if (mx) hipLaunchKernelGGL((vgl_k_reduce_partials<MODE, true, R>), dim3(nb), dim3(VGL_BLOCK), 0, st, n, v.rowptr, f.get_flags(), f.get_ids(), reduce_op, reduce_partials); \
            else hipLaunchKernelGGL((vgl_k_reduce_partials<MODE, false, R>), dim3(nb), dim3(VGL_BLOCK), 0, st, n, v.rowptr, f.get_flags(), f.get_ids(), reduce_op, reduce_partials); \
        } while (0)
        if (t == ALL_ACTIVE_FRONTIER) VGL_REDUCE_LAUNCH(0);
        else if (t == DENSE_FRONTIER) VGL_REDUCE_LAUNCH(1);
        else VGL_REDUCE_LAUNCH(2);
#undef VGL_REDUCE_LAUNCH
        VGL_HIP_RT(hipGetLastError());
        double r = 0.0;
        if (mx) {
            hipLaunchKernelGGL(vgl_k_max_fold, dim3(1), dim3(VGL_BLOCK), 0, st, nb, (const double *)reduce_partials, reduce_partials + 1024);
            VGL_HIP_RT(hipGetLastError());
            VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), &r, reduce_partials + 1024, sizeof(double)));
        } else
            VGL_HIP_CALL(vgl_hip_reduce_sum_f64_buffer(VGL_RUNTIME::ctx(), nb, reduce_partials, &r));      // fixed-order fold of the partials
        performance_stats.update_reduce_stats(watch.seconds(), (size_t)n);
        return (_T)r;
    }

    template <typename FilterCondition>
    void generate_new_frontier(VGL_Graph &g, VGL_Frontier &f, FilterCondition &&filter_cond)
    {
        const vgl_csr_view v = g.get_direction_view(current_traversal_direction);
        using C = typename std::decay<FilterCondition>::type;
        const int V = g.get_vertices_count();
        const vgl_stopwatch watch;
        f.set_direction(current_traversal_direction);
        // CSR_GRAPH frontiers are ALL_ACTIVE or SPARSE (generate_new_frontier.hpp:113-164); VECTOR_CSR_GRAPH ones turn DENSE (flags only)
        // above FRONTIER_TYPE_CHANGE_THRESHOLD = 0.7 of the vertices (generate_new_frontier.hpp:67-91, settings.h)
        const double dense_threshold = g.get_format() == VECTOR_CSR_GRAPH ? 0.7 : 0.0;
        // ONE pass evaluates the condition, writes its bits (a V / 8-byte bitmap; int32 flags only under VGL_GNF_INT_FLAGS=1) and counts (the last
        // workgroup scans the per-tile counts and hands size and neighbour count to the host); the compaction then reads the bitmap and leaves the
        // ids' out-edge offsets behind for scatter; a DENSE / ALL_ACTIVE result gets its int32 flags from the bitmap
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        const bool plan = current_traversal_direction == SCATTER;
        vgl_hip_gnf_buffers b;
        VGL_HIP_CALL(vgl_hip_gnf_begin(c, g.get_handle(), f.get_handle(), plan ? 1 : 0, &b));
        const vgl_pred_user<C> pred{filter_cond, v.rowptr};
        hipLaunchKernelGGL((vgl_k_gnf_count<vgl_pred_user<C>>), dim3((unsigned)b.nvtiles), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), pred, b.nrows, b.row_begin,
                           b.out_rowptr, b.vt_cnt, b.vt_deg, b.front_bytes, (uint8_t *)nullptr, b.flags, b.ticket, b.vt_cnt_off, b.vt_deg_off, b.counters,
                           b.plan_offs, b.host_counters, b.seq);
        VGL_HIP_RT(hipGetLastError());
        VGL_HIP_CALL(vgl_hip_gnf_complete(c, g.get_handle(), f.get_handle(), dense_threshold, plan ? 1 : 0, b.seq));
        performance_stats.update_gnf_stats(watch.seconds(), (size_t)V);
    }

    // ---- exchange_vertices_array (common/graph_abstractions.h:157-168, common/mpi_exchange.hpp:300-365): the three policies of the
    //      reference over the library's communicator (RCCL on the context's stream).  merge_op must be callable on the device
    //      ([] __device__ (T received, T mine) -> T); a world of one returns at once, like the reference. ----
    template <typename _TGraph, typename _T>
    void exchange_vertices_array(DataExchangePolicy policy, _TGraph &g, VerticesArray<_T> &data)
    {
        if (VGL_RUNTIME::get_mpi_proc_num() == 1) return;
        if (policy == EXCHANGE_RECENTLY_CHANGED)
            throw "Error in GraphAbstractionsHIP::exchange_vertices_array : old data must be provided for EXCHANGE_RECENTLY_CHANGED";
        if (policy != EXCHANGE_PRIVATE_DATA) throw "Currently not supported";
        // every rank computed the entries of its own vertex range: all-gather of the owned slices in place (exchange_data_private)
        const std::vector<int64_t> &bounds = g.get_mpi_bounds(current_traversal_direction);
        VGL_HIP_CALL(vgl_hip_exchange_allgather_slices(VGL_RUNTIME::comm(), data.get_ptr(), bounds.data(), (int)sizeof(_T)));
        VGL_RUNTIME::sync();
    }
    template <typename _TGraph, typename _T, typename MergeOp>
    void exchange_vertices_array(DataExchangePolicy policy, _TGraph &g, VerticesArray<_T> &data, MergeOp &&merge_op)
    {
        if (VGL_RUNTIME::get_mpi_proc_num() == 1) return;
        if (policy != EXCHANGE_ALL)
            throw "Error in GraphAbstractionsHIP::exchange_vertices_array : old data is NOT provided for NON EXCHANGE_RECENTLY_CHANGED";
        exchange_all(data, merge_op);
        VGL_RUNTIME::sync();
    }
    template <typename _TGraph, typename _T, typename MergeOp>
    void exchange_vertices_array(DataExchangePolicy policy, _TGraph &g, VerticesArray<_T> &data, VerticesArray<_T> &old_data, MergeOp &&merge_op)
    {
        if (VGL_RUNTIME::get_mpi_proc_num() == 1) return;
        if (policy != EXCHANGE_RECENTLY_CHANGED)
            throw "Error in GraphAbstractionsHIP::exchange_vertices_array : old data is provided for NON EXCHANGE_RECENTLY_CHANGED";
        exchange_changed(data, old_data, merge_op);
        VGL_RUNTIME::sync();
    }

private:
    void *exchange_scratch = nullptr; size_t exchange_scratch_bytes = 0; int *exchange_heads = nullptr;
    void *exchange_buffer(size_t bytes)
    {
        if (exchange_scratch_bytes < bytes) {
            MemoryAPI::free_device_array((char *)exchange_scratch);
            exchange_scratch = nullptr; exchange_scratch_bytes = 0;
            char *p = nullptr;
            MemoryAPI::allocate_device_array(&p, bytes);
            exchange_scratch = p; exchange_scratch_bytes = bytes;
        }
        return exchange_scratch;
    }
    // EXCHANGE_ALL: every rank receives every copy and folds them with merge_op in rank order
    template <typename _T, typename MergeOp>
    void exchange_all(VerticesArray<_T> &data, MergeOp &&merge_op)
    {
        const int P = VGL_RUNTIME::get_mpi_proc_num(), rank = VGL_RUNTIME::get_mpi_rank(), n = data.size();
        _T *all = (_T *)exchange_buffer(sizeof(_T) * (size_t)n * (size_t)P);
        VGL_HIP_CALL(vgl_hip_exchange_allgather(VGL_RUNTIME::comm(), data.get_ptr(), all, (int64_t)sizeof(_T) * n));
        using M = typename std::decay<MergeOp>::type;
        hipLaunchKernelGGL((vgl_k_merge_copies<_T, M>), dim3(grid_for(n)), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), n, P, rank, (const _T *)all, data.get_ptr(), merge_op);
        VGL_HIP_RT(hipGetLastError());
    }
    // EXCHANGE_RECENTLY_CHANGED: (index, value) pairs of the entries that differ from old_data, from every rank; the heads of the lists
    // (count + 2048 pairs) travel first and are the whole exchange when nobody changed more; beyond n / (2 ranks) changed entries on some
    // rank the whole arrays are exchanged instead (fewer bytes than the lists)
    template <typename _T, typename MergeOp>
    typename std::enable_if<sizeof(_T) == 4>::type exchange_changed(VerticesArray<_T> &data, VerticesArray<_T> &old_data, MergeOp &&merge_op)
    {
        const int P = VGL_RUNTIME::get_mpi_proc_num(), rank = VGL_RUNTIME::get_mpi_rank(), n = data.size();
        constexpr long long SMALL = 2048;
        const long long cap = std::max<long long>(SMALL, n / (2 * P)), small_stride = 1 + 2 * SMALL;
        const size_t mine_ints = (size_t)(1 + 2 * cap);
        int *mine = (int *)exchange_buffer(sizeof(int) * (mine_ints + (size_t)P * mine_ints));
        int *all = mine + mine_ints;
        vgl_hip_ctx *c = VGL_RUNTIME::ctx();
        if (!exchange_heads) MemoryAPI::allocate_array(&exchange_heads, 64);
        VGL_HIP_CALL(vgl_hip_diff_to_pairs_u32(c, n, old_data.get_ptr(), data.get_ptr(), (int32_t)cap, mine));
        VGL_HIP_CALL(vgl_hip_exchange_allgather(VGL_RUNTIME::comm(), mine, all, small_stride * 4));
        hipLaunchKernelGGL(vgl_k_list_heads, dim3(1), dim3(64), 0, VGL_RUNTIME::stream(), (const int *)all, small_stride, P, exchange_heads);
        VGL_HIP_RT(hipGetLastError());
        VGL_RUNTIME::sync();
        long long most = 0;
        for (int p = 0; p < P; p++) most = std::max<long long>(most, exchange_heads[p]);
        if (most == 0) return;
        if (most > cap) { exchange_all(data, merge_op); return; }
        long long stride = small_stride, pairs = SMALL;
        if (most > SMALL) {
            pairs = 1;
            while (pairs < most) pairs <<= 1;
            pairs = std::min(pairs, cap);
            stride = 1 + 2 * pairs;
            VGL_HIP_CALL(vgl_hip_exchange_allgather(VGL_RUNTIME::comm(), mine, all, stride * 4));
        }
        using M = typename std::decay<MergeOp>::type;
        for (int p = 0; p < P; p++)
            if (p != rank && exchange_heads[p] > 0)
                hipLaunchKernelGGL((vgl_k_merge_pairs<_T, M>), dim3(grid_for(exchange_heads[p])), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), (const int *)(all + (size_t)p * stride),
                                   pairs, n, data.get_ptr(), merge_op);
        VGL_HIP_RT(hipGetLastError());
    }
    template <typename _T, typename MergeOp>
    typename std::enable_if<sizeof(_T) != 4>::type exchange_changed(VerticesArray<_T> &data, VerticesArray<_T> &, MergeOp &&merge_op)
    {
        exchange_all(data, merge_op);       // pair lists carry 4-byte values: wider types take the whole-array exchange
    }

public:
    // advance with one lane per active vertex, edges in adjacency order between pre and post (vgl_k_advance_rows): the reference's vector-core shape
    void enable_sequential_rows() { sequential_rows = true; }
    void disable_sequential_rows() { sequential_rows = false; }
    void enable_safe_stores() {}         // no-op off NEC (graph_abstractions_multicore.h:295-296)
    void disable_safe_stores() {}
};
