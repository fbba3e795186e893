// graph_abstractions_hip.h -- the MI355X backend class ON THE REFERENCE'S OWN CONTAINERS.  This file is what a VGL maintainer adds as
// vgl_compute_api/hip/graph_abstractions_hip.h (integration/apply_hip_binding.py copies it there and makes the few edits of INTEGRATION.md
// section 2 to the surrounding files): a class derived from GraphAbstractions, selected by -D __USE_HIP__ through VGL_GRAPH_ABSTRACTIONS
// (architecture_independent_api.h), with the worker set the common dispatch layer calls (common/advance.hpp, compute.hpp, reduce.hpp,
// generate_new_frontier.hpp) -- the shape of vgl_compute_api/template/graph_abstractions_template.h:5-107 and of the CUDA backend
// vgl_compute_api/gpu/graph_abstractions_gpu.h:17-190.  It reads CSRGraph / VectorCSRGraph through their public accessors and writes
// FrontierCSR / FrontierVectorCSR through friend access, exactly like GraphAbstractionsGPU / GraphAbstractionsMulticore.
//
//   kernels   : vectorgraphlibrary_amd/hip/vgl_hip_kernels.hpp (templated on the user's device lambdas; plain pointers only)
//   library   : libvgl_hip.so through the C ABI include/vgl_hip.h -- graph handles work on DEVICE COPIES of the containers' vertex_pointers / adjacent_ids
//               (+ the vector extension), made when a container is first used; frontier handles BORROW the containers' flags / ids
//               (vgl_hip_frontier_create_on), so host code of the reference that writes those arrays (add_vertex, set_all_active) needs no change
//   memory    : the containers stay in managed memory (MemoryAPI::allocate_array = hipMallocManaged under __USE_HIP__: host-resident on a pool
//               without XNACK, where the reference's host code wants them); user arrays (VerticesArray / EdgesArray) are shadowed -- a pinned host
//               mirror and a buffer in HBM with an owner flag, shadow_memory.h: every primitive starts with hip_shadows_to_device(), which uploads what
//               host code wrote since the last one; frontier flags / ids are device memory
//
// Containers served: CSR_GRAPH (advance_worker.hpp:62-149, generate_new_frontier.hpp:113-164) and VECTOR_CSR_GRAPH with its three degree
// ranges and the padded vector extension (advance_worker.hpp:204-319, advance_{all_active,dense,sparse}.hpp, generate_new_frontier.hpp:29-111):
// the collective range of an ALL_ACTIVE or DENSE frontier walks the VE copy of the adjacency and hands the collective operators VE-space
// global_edge_pos (process shift of VE_STORAGE + segment start + edge_pos * VECTOR_LENGTH + lane), a SPARSE frontier hands them CSR positions,
// as the reference does.  EDGES_LIST_GRAPH (round 5; the condensed graph of TransitiveClosure::vgl_purdoms, tc.hpp:113-133): every edge of the list per
// advance, flags + count per frontier generation, as gpu/advance.hpp:44-68 and gpu/generate_new_frontier.hpp:266-292.  CSR_VG_GRAPH throws (as the CUDA
// backend does for what it lacks, gpu/advance.hpp:37).
// Every primitive returns synchronised (SAFE_KERNEL_CALL of the CUDA backend, cuda_error_handling.h:15-27).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <type_traits>
#include "vgl_hip_kernels.hpp"          // -I <repository>/vectorgraphlibrary_amd/hip
#include "vector_register/vector_registers.h"

#define VGL_HIP_BIND_CALL(expr) do { if ((expr) != 0) throw vgl_hip_last_error(); } while (0)
#define VGL_HIP_BIND_RT(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw hipGetErrorString(_e); } while (0)

// the collective range of a VECTOR_CSR_GRAPH over its vector extension: one wavefront per segment of VECTOR_LENGTH = 64 vertices, lane i owns vertex
// first + i and walks its entries in order between its pre and post operators (advance_all_active.hpp:134-225, advance_dense.hpp:137-235); the
// loads of ve_adjacent_ids are one coalesced 256-byte row per step.  DENSE: flagged vertices only -- their edges AND their pre / post operators.
// (The reference's dense collective kernel calls pre / post for every vertex of the range, flagged or not, advance_dense.hpp:161-166,226-232.  That
// quirk is dropped on purpose: Coloring::vgl_coloring's post operator assigns a colour from a mask that only an ACTIVE vertex's edges have filled,
// so on a DENSE frontier it would recolour every settled vertex with the first colour of the range -- seen here as 7766 conflicting edges on
// uniform-11 x 16 in vcsr.)
template <bool DENSE, class EdgeOp, class PreOp, class PostOp>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_advance_vector_extension(int segments, int starting_vertex, int vertices_count, const long long *vertex_pointers,
                                                                            const long long *group_ptrs, const int *group_sizes, const int *ve_adjacent_ids,
                                                                            const int *flags, long long process_shift, EdgeOp edge_op, PreOp pre_op, PostOp post_op)
{
    const int segment = (int)((blockIdx.x * (unsigned)VGL_BLOCK + threadIdx.x) >> 6);
    const int lane = (int)(threadIdx.x & 63);
    if (segment >= segments) return;
    const int src_id = segment * 64 + starting_vertex + lane;
    const long long segment_edges_start = group_ptrs[segment];
    const int segment_connections_count = group_sizes[segment];
    const bool present = src_id < vertices_count;
    int connections_count = 0;
    if (present && segment_connections_count > 0) connections_count = (int)(vertex_pointers[src_id + 1] - vertex_pointers[src_id]);
    const bool walks = present && (!DENSE || flags[src_id] > 0);
    if (walks) pre_op(src_id, connections_count, lane);
    for (int edge_pos = 0; edge_pos < segment_connections_count; edge_pos++) {
        const long long internal_edge_pos = segment_edges_start + (long long)edge_pos * 64 + lane;
        if (walks && edge_pos < connections_count) edge_op(src_id, ve_adjacent_ids[internal_edge_pos], edge_pos, process_shift + internal_edge_pos, lane);
    }
    if (walks) post_op(src_id, connections_count, lane);
}
// sizes and degree sums of the three parts of a FrontierVectorCSR (estimate_sorted_frontier_part_size, generate_new_frontier.hpp:3-27) from what the
// count pass of the frontier generation left: per 2048-vertex tile the number of flagged vertices and their degree sum (vt_cnt / vt_deg), and the
// predicate's bitmap for the (at most two) tiles a part boundary cuts.  ONE workgroup, microseconds -- the pass over all V int32 flags and their row
// offsets that this replaces cost 55 us per BFS level on RMAT-24 (profiles/r05_binding_bfs_vcsr_rmat24_kernel_stats.csv), a twelfth of a traversal.
constexpr int VGL_PARTS_THREADS = 1024;
__global__ __launch_bounds__(VGL_PARTS_THREADS) void vgl_k_frontier_parts_from_tiles(int vertices_count, long long tiles, const int *vt_cnt, const long long *vt_deg,
                                                                                     const unsigned char *front_bytes, const long long *vertex_pointers, int ve_threshold,
                                                                                     int vc_threshold, unsigned long long *out)      // out[0..2] sizes, out[3..5] neighbours
{
    unsigned long long n[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    const int bounds[4] = {0, ve_threshold, vc_threshold, vertices_count};
    for (long long t = threadIdx.x; t < tiles; t += VGL_PARTS_THREADS) {
        const long long first = t * VGL_TILE, last = min((long long)vertices_count, first + VGL_TILE);      // [first, last)
        for (int p = 0; p < 3; p++)
            if (first >= bounds[p] && last <= bounds[p + 1]) { n[p] += (unsigned long long)vt_cnt[t]; d[p] += (unsigned long long)vt_deg[t]; }
    }
    // the tiles cut by a boundary, vertex by vertex (all threads, both boundaries)
    for (int b = 1; b <= 2; b++) {
        const int cut = bounds[b];
        if (cut <= 0 || cut >= vertices_count || cut % VGL_TILE == 0) continue;
        if (b == 2 && bounds[1] > 0 && bounds[1] / VGL_TILE == cut / VGL_TILE && bounds[1] % VGL_TILE != 0) continue;     // same tile as the first cut: done there
        const long long first = (long long)(cut / VGL_TILE) * VGL_TILE, last = min((long long)vertices_count, first + VGL_TILE);
        for (long long v = first + threadIdx.x; v < last; v += VGL_PARTS_THREADS)
            if ((front_bytes[v >> 3] >> (v & 7)) & 1) {
                const int part = v < ve_threshold ? 0 : (v < vc_threshold ? 1 : 2);
                n[part] += 1; d[part] += (unsigned long long)(vertex_pointers[v + 1] - vertex_pointers[v]);
            }
    }
    __shared__ unsigned long long s_part[VGL_PARTS_THREADS / 64][6];
    for (int p = 0; p < 3; p++) {
        for (int o = 32; o > 0; o >>= 1) { n[p] += __shfl_xor(n[p], o); d[p] += __shfl_xor(d[p], o); }
        if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6][p] = n[p]; s_part[threadIdx.x >> 6][3 + p] = d[p]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long t = 0;
        for (int w = 0; w < VGL_PARTS_THREADS / 64; w++) t += s_part[w][threadIdx.x];
        out[threadIdx.x] = t;
        __threadfence_system();                              // (the words live in pinned host memory)
    }
}
// SPARSE frontier of a VECTOR_CSR_GRAPH: ids below the vector-core threshold take (edge_op, pre, post), the others the collective set; ids are
// degree-sorted positions, so the classes ARE id ranges (vect_csr_graph.h:40-41)
template <class A, class B>
struct vgl_range_edge_op {
    A first; B collective; int threshold;
    __device__ __forceinline__ void operator()(int src, int dst, int local, long long global, int lane) const
    {
        if (src < threshold) first(src, dst, local, global, lane);
        else collective(src, dst, local, global, lane);
    }
};
template <class A, class B>
struct vgl_range_vertex_op {
    A first; B collective; int threshold;
    __device__ __forceinline__ void operator()(int src, int connections, int lane) const
    {
        if (src < threshold) first(src, connections, lane);
        else collective(src, connections, lane);
    }
};

// EDGES_LIST_GRAPH: one lane per stored edge (multicore/advance_worker.hpp:10-57, gpu/advance.hpp:44-68: the frontier is not consulted, local and
// global edge position are both the position in the list).  Endpoints outside [0, vertices_count) are skipped: tc.hpp:104-110 sizes its condensed
// graph by the LARGEST component id (one vertex short; the binding adds the missing one, apply_hip_binding.py) -- a kernel must not follow such an id.
template <class EdgeOp>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_advance_edges_list(long long edges_count, int vertices_count, const int *src_ids, const int *dst_ids, EdgeOp edge_op)
{
    for (long long e = (long long)blockIdx.x * VGL_BLOCK + threadIdx.x; e < edges_count; e += (long long)gridDim.x * VGL_BLOCK) {
        const int src_id = src_ids[e], dst_id = dst_ids[e];
        if ((unsigned)src_id < (unsigned)vertices_count && (unsigned)dst_id < (unsigned)vertices_count)
            edge_op(src_id, dst_id, (int)e, e, (int)(threadIdx.x & 63));
    }
}
// generate_new_frontier on an EDGES_LIST_GRAPH (multicore/generate_new_frontier.hpp:235-270): flags from the condition (connections_count = 0, as there),
// their count in *count
template <class Cond>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_edges_list_flags(int vertices_count, Cond cond, int *flags, unsigned long long *count)
{
    __shared__ unsigned long long s_count[VGL_BLOCK / 64];
    unsigned long long n = 0;
    for (int v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < vertices_count; v += gridDim.x * VGL_BLOCK) {
        const int flag = cond(v, 0);
        flags[v] = flag;
        n += flag > 0;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) s_count[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < VGL_BLOCK / 64; w++) n += s_count[w];
        if (n) atomicAdd(count, n);                        // one atomic per workgroup
    }
}

class GraphAbstractionsHIP : public GraphAbstractions
{
private:
    vgl_hip_ctx *ctx;
    hipStream_t stream;
    double *reduce_partials;                        // 1024 partials + the folded maximum
    unsigned long long *part_counters;              // device: the edges-list frontier generation's counter (atomics on host-resident memory would cross PCIe one by one)
    unsigned long long *part_sizes;                 // PINNED host memory: vgl_k_frontier_parts_from_tiles stores its six results here (a hipMemcpyAsync into a stack
                                                    // array took 22 - 58 us per BFS level in the API trace: pageable destination, staged copy)
    // Per direction container: the library's graph handle and DEVICE copies of the arrays the kernels traverse.  The containers themselves stay in
    // managed memory (host-resident on a pool without XNACK), where the reference's host code -- import, select_random_nz_vertex, the sequential
    // checkers -- reads them at full speed; a container that was resized / re-imported (other arrays or another edge count behind the same object)
    // gets new copies.
    struct graph_binding {
        vgl_hip_graph *handle = nullptr;
        const void *vertex_pointers = nullptr, *adjacent_ids = nullptr;       // the container's arrays the copies were made from
        long long edges_count = 0;
        unsigned long long version = 0;                                       // hip_container_version() when the copies were made (bumped by the container's free())
        long long *d_vertex_pointers = nullptr;
        int *d_adjacent_ids = nullptr;
        long long *d_ve_group_ptrs = nullptr;                                   // VECTOR_CSR_GRAPH: the vector extension
        int *d_ve_group_sizes = nullptr, *d_ve_adjacent_ids = nullptr;
    };
    // The bindings outlive the backend object: the reference's algorithms construct a VGL_GRAPH_ABSTRACTIONS per call (bfs.hpp:58, gpu_pr.hpp:15),
    // and copying a container's adjacency per call would cost more than the traversal (RMAT-20: 3.7 of 4.4 ms per BFS before this was a
    // process-wide table).  Entries are replaced when the container's arrays change (other pointers or another edge count behind the same object)
    // and released at process exit; host code that REWRITES a container's adjacency in place after the first primitive -- nothing in the reference
    // does -- calls forget_graph_bindings().
    static std::map<void *, graph_binding> &graph_bindings() { static std::map<void *, graph_binding> table; return table; }
    // every release of a binding starts a new generation: frontier handles of ANY backend object that were made before it refer to a graph handle
    // that may be gone and are dropped (unread) at their next use
    static unsigned long long &bindings_generation() { static unsigned long long generation = 0; return generation; }
    unsigned long long frontier_generation = 0;
    // enable_safe_stores() (common/graph_abstractions.h:170; coloring.hpp:108-110 brackets a scatter whose operator does a plain read-modify-write of
    // src-indexed data with it): while on, an advance runs ONE LANE PER ACTIVE VERTEX -- its edges in adjacency order between its pre and post
    // operators, the execution shape of the reference's vector-core kernels (multicore/advance_worker.hpp:62-149) -- instead of edge tiles
    bool safe_stores = false;
    // one per frontier container (its flags / ids are borrowed).  plan_token / planned_on: generate_new_frontier left the advance plan of the SPARSE
    // frontier behind (the ids' edge offsets, a by-product of the compaction) and stamped the container (BaseFrontier::hip_plan_token); the container's
    // mutators -- set_all_active, add_vertex, clear -- zero the stamp, so an equal non-zero stamp says that nothing touched the frontier since and
    // scatter / gather on the same direction container start from the plan instead of three more launches and a host wait per level
    struct frontier_binding { vgl_hip_frontier *handle = nullptr; unsigned long long plan_token = 0; vgl_hip_graph *planned_on = nullptr; int planned_size = 0; };
    std::map<void *, frontier_binding> frontier_handles;
    static unsigned long long next_plan_token() { static unsigned long long token = 0; return ++token; }
    void drop_frontier_handles()
    {
        for (auto &kv : frontier_handles) vgl_hip_frontier_destroy(ctx, kv.second.handle);
        frontier_handles.clear();
    }
    static int grid_for(long long n) { return (int)std::min<long long>(4096, std::max<long long>(1, (n + VGL_BLOCK - 1) / VGL_BLOCK)); }
    static int sparsity_code(FrontierSparsityType t)
    { return t == ALL_ACTIVE_FRONTIER ? VGL_HIP_FRONTIER_ALL_ACTIVE : (t == DENSE_FRONTIER ? VGL_HIP_FRONTIER_DENSE : VGL_HIP_FRONTIER_SPARSE); }
    void finish() { VGL_HIP_BIND_RT(hipGetLastError()); VGL_HIP_BIND_RT(hipStreamSynchronize(stream)); }

    template <typename T>
    T *device_copy(const T *src, size_t n)
    {
        T *d = nullptr;
        VGL_HIP_BIND_RT(hipMalloc((void **)&d, sizeof(T) * std::max<size_t>(n, 1)));
        if (n) VGL_HIP_BIND_RT(hipMemcpyAsync(d, src, sizeof(T) * n, hipMemcpyDefault, stream));
        return d;
    }
    void release(graph_binding &b)
    {
        if (b.handle) vgl_hip_graph_destroy(ctx, b.handle);
        hipFree(b.d_vertex_pointers); hipFree(b.d_adjacent_ids); hipFree(b.d_ve_group_ptrs); hipFree(b.d_ve_group_sizes); hipFree(b.d_ve_adjacent_ids);
        b = graph_binding();
    }
    template <typename GraphContainer>
    const graph_binding &binding_of(GraphContainer &_graph)
    {
        std::map<void *, graph_binding> &graph_handles = graph_bindings();
        auto it = graph_handles.find((void *)&_graph);
        if (it != graph_handles.end()) {
            graph_binding &b = it->second;
            if (b.vertex_pointers == (const void *)_graph.get_vertex_pointers() && b.adjacent_ids == (const void *)_graph.get_adjacent_ids() &&
                b.edges_count == (long long)_graph.get_edges_count() && b.version == hip_container_version((const void *)&_graph))
                return b;
            drop_frontier_handles();                                                         // (they refer to the old handle)
            release(b);
            graph_handles.erase(it);
            frontier_generation = ++bindings_generation();
        }
        graph_binding b;
        const size_t V = (size_t)_graph.get_vertices_count(), E = (size_t)_graph.get_edges_count();
        b.vertex_pointers = (const void *)_graph.get_vertex_pointers(); b.adjacent_ids = (const void *)_graph.get_adjacent_ids(); b.edges_count = (long long)E;
        b.version = hip_container_version((const void *)&_graph);
        b.d_vertex_pointers = device_copy(_graph.get_vertex_pointers(), V + 1);
        b.d_adjacent_ids = device_copy(_graph.get_adjacent_ids(), E);
        copy_vector_extension(_graph, b);
        VGL_HIP_BIND_CALL(vgl_hip_graph_create(ctx, (int32_t)V, 0, (int32_t)V, (const int64_t *)b.d_vertex_pointers, b.d_adjacent_ids, (int64_t)E, nullptr, nullptr, 0, &b.handle));
        return graph_handles[(void *)&_graph] = b;
    }
    void copy_vector_extension(CSRGraph &, graph_binding &) {}
    void copy_vector_extension(VectorCSRGraph &_graph, graph_binding &b)
    {
        VectorExtension *ve = _graph.get_ve_ptr();
        const size_t segments = (size_t)ve->get_vector_segments_count();
        b.d_ve_group_ptrs = device_copy(ve->get_vector_group_ptrs(), segments);
        b.d_ve_group_sizes = device_copy(ve->get_vector_group_sizes(), segments);
        b.d_ve_adjacent_ids = device_copy(ve->get_adjacent_ids(), (size_t)ve->get_edges_count_in_ve());
    }
    template <typename GraphContainer>
    vgl_hip_graph *handle_of(GraphContainer &_graph) { return binding_of(_graph).handle; }

    // EDGES_LIST_GRAPH: device copies of the two id arrays and an exclusive prefix of the containers' per-vertex edge counts (get_connections_count) in
    // the place of vertex_pointers, so that the per-vertex kernels hand the operators the connection counts the reference's do.  Same identity rule as
    // the CSR bindings (EdgesListGraph::free bumps the version).
    struct edges_list_binding {
        const void *src_ids = nullptr, *dst_ids = nullptr;
        long long edges_count = 0;
        int vertices_count = 0;
        unsigned long long version = 0;
        int *d_src_ids = nullptr, *d_dst_ids = nullptr;
        long long *d_degree_prefix = nullptr;
    };
    static std::map<void *, edges_list_binding> &edges_list_bindings() { static std::map<void *, edges_list_binding> table; return table; }
    const edges_list_binding &edges_list_of(EdgesListGraph &_graph)
    {
        std::map<void *, edges_list_binding> &table = edges_list_bindings();
        auto it = table.find((void *)&_graph);
        if (it != table.end()) {
            edges_list_binding &b = it->second;
            if (b.src_ids == (const void *)_graph.get_src_ids() && b.dst_ids == (const void *)_graph.get_dst_ids() && b.edges_count == (long long)_graph.get_edges_count() &&
                b.vertices_count == _graph.get_vertices_count() && b.version == hip_container_version((const void *)&_graph))
                return b;
            hipStreamSynchronize(stream);
            hipFree(b.d_src_ids); hipFree(b.d_dst_ids); hipFree(b.d_degree_prefix);
            table.erase(it);
        }
        edges_list_binding b;
        const size_t V = (size_t)_graph.get_vertices_count(), E = (size_t)_graph.get_edges_count();
        b.src_ids = (const void *)_graph.get_src_ids(); b.dst_ids = (const void *)_graph.get_dst_ids(); b.edges_count = (long long)E; b.vertices_count = (int)V;
        b.version = hip_container_version((const void *)&_graph);
        b.d_src_ids = device_copy(_graph.get_src_ids(), E);
        b.d_dst_ids = device_copy(_graph.get_dst_ids(), E);
        std::vector<long long> prefix(V + 1, 0);
        for (size_t v = 0; v < V; v++) prefix[v + 1] = prefix[v] + _graph.get_connections_count((int)v);
        b.d_degree_prefix = device_copy(prefix.data(), V + 1);
        VGL_HIP_BIND_RT(hipStreamSynchronize(stream));                                   // (prefix is a local)
        return table[(void *)&_graph] = b;
    }

    // the frontier container may have been changed by host code since the last primitive (add_vertex, clear, set_all_active write its fields and
    // arrays directly): its description is taken as it stands before every use
    template <typename FrontierContainer>
    vgl_hip_frontier *handle_of(FrontierContainer &_frontier, vgl_hip_graph *_graph_handle)
    {
        vgl_hip_frontier *h = nullptr;
        if (frontier_generation != bindings_generation()) { drop_frontier_handles(); frontier_generation = bindings_generation(); }
        auto it = frontier_handles.find((void *)&_frontier);
        if (it != frontier_handles.end() && vgl_hip_frontier_flags(it->second.handle) != _frontier.flags) {     // another frontier object at a recycled address
            vgl_hip_frontier_destroy(ctx, it->second.handle);
            frontier_handles.erase(it);
            it = frontier_handles.end();
        }
        if (it != frontier_handles.end()) {
            frontier_binding &fb = it->second;
            if (fb.plan_token != 0 && fb.plan_token == _frontier.hip_plan_token && fb.planned_on == _graph_handle && fb.planned_size == _frontier.size &&
                _frontier.sparsity_type == SPARSE_FRONTIER)
                return fb.handle;                                                   // as generated: description and plan stand
            fb.plan_token = 0;
            h = fb.handle;
        } else {
            VGL_HIP_BIND_CALL(vgl_hip_frontier_create_on(ctx, _graph_handle, _frontier.flags, _frontier.ids, &h));
            frontier_handles[(void *)&_frontier].handle = h;
        }
        VGL_HIP_BIND_CALL(vgl_hip_frontier_set_state(ctx, h, _graph_handle, _frontier.size, _frontier.neighbours_count, sparsity_code(_frontier.sparsity_type)));
        return h;
    }
    // generate_new_frontier produced _frontier on _graph_handle: a SPARSE one carries its advance plan from here on
    template <typename FrontierContainer>
    void stamp_generated(FrontierContainer &_frontier, vgl_hip_graph *_graph_handle, int _sparsity)
    {
        frontier_binding &fb = frontier_handles[(void *)&_frontier];
        fb.plan_token = _sparsity == VGL_HIP_FRONTIER_SPARSE ? next_plan_token() : 0;
        fb.planned_on = _graph_handle;
        fb.planned_size = _frontier.size;
        _frontier.hip_plan_token = fb.plan_token;
    }

    // per-vertex operator over the active vertices of a range of ids
    template <typename Op>
    void vertex_pass(int _vertices_count, const long long *_vertex_pointers, FrontierSparsityType _type, int *_flags, int *_ids, int _frontier_size,
                     int _row_lo, int _row_hi, Op &&op)
    {
        using O = typename std::decay<Op>::type;
        if (_type == ALL_ACTIVE_FRONTIER)
            hipLaunchKernelGGL((vgl_k_vertex_op<0, O>), dim3(grid_for(_vertices_count)), dim3(VGL_BLOCK), 0, stream, _vertices_count, _vertex_pointers, _flags, _ids, _row_lo, _row_hi, op);
        else if (_type == DENSE_FRONTIER)
            hipLaunchKernelGGL((vgl_k_vertex_op<1, O>), dim3(grid_for(_vertices_count)), dim3(VGL_BLOCK), 0, stream, _vertices_count, _vertex_pointers, _flags, _ids, _row_lo, _row_hi, op);
        else if (_frontier_size > 0)
            hipLaunchKernelGGL((vgl_k_vertex_op<2, O>), dim3(grid_for(_frontier_size)), dim3(VGL_BLOCK), 0, stream, _frontier_size, _vertex_pointers, _flags, _ids, _row_lo, _row_hi, op);
    }
    template <typename Op> static constexpr bool is_empty_op() { return std::is_empty<typename std::decay<Op>::type>::value && std::is_same<typename std::decay<Op>::type, decltype(EMPTY_VERTEX_OP)>::value; }

    // edges of the active vertices with ids in [_row_lo, _row_hi): static edge tiles (ALL_ACTIVE / DENSE) or the frontier's own edge space (SPARSE)
    template <typename GraphContainer, typename FrontierContainer, typename EdgeOp>
    void edge_pass(GraphContainer &_graph, FrontierContainer &_frontier, long long _process_shift, int _row_lo, int _row_hi, EdgeOp &&edge_op)
    {
        using E = typename std::decay<EdgeOp>::type;
        LOAD_FRONTIER_DATA(_frontier);
        const graph_binding &gb = binding_of(_graph);
        const long long *host_vertex_pointers = _graph.get_vertex_pointers();      // (read on the host below: the container's own array)
        const long long *vertex_pointers = gb.d_vertex_pointers;
        const int *adjacent_ids = gb.d_adjacent_ids;
        const long long edges_count = _graph.get_edges_count();
        vgl_hip_graph *gh = gb.handle;
        if (_frontier.get_sparsity_type() == SPARSE_FRONTIER) {
            if (frontier_size == 0) return;
            vgl_hip_frontier *fh = handle_of(_frontier, gh);
            const int64_t *offs; const int32_t *tile_first; int64_t M;
            VGL_HIP_BIND_CALL(vgl_hip_frontier_advance_plan(ctx, gh, fh, 0, &offs, &tile_first, &M));
            if (M > 0)
                hipLaunchKernelGGL((vgl_k_advance_sparse<E>), dim3((unsigned)((M + VGL_TILE - 1) / VGL_TILE)), dim3(VGL_ADV_THREADS), 0, stream, frontier_ids, offs, tile_first,
                                   frontier_size, (long long)M, vertex_pointers, adjacent_ids, _process_shift, _row_lo, _row_hi, edge_op);
            return;
        }
        if (edges_count == 0 || _row_hi <= _row_lo) return;
        const int32_t *tile_row; int64_t ntiles;
        VGL_HIP_BIND_CALL(vgl_hip_graph_tile_rows(gh, 0, &tile_row, &ntiles));
        // rows are stored in id order: the tiles past the last edge of row _row_hi - 1 hold nothing of the range
        const long long last_edge = host_vertex_pointers[_row_hi];
        const long long first_edge = host_vertex_pointers[_row_lo];
        if (last_edge <= first_edge) return;
        const unsigned tiles = (unsigned)std::min<long long>(ntiles, (last_edge + VGL_TILE - 1) / VGL_TILE);
        if (_frontier.get_sparsity_type() == DENSE_FRONTIER)
            hipLaunchKernelGGL((vgl_k_advance_static<true, E>), dim3(tiles), dim3(VGL_BLOCK), 0, stream, vertex_pointers, adjacent_ids, tile_row, edges_count, _process_shift,
                               frontier_flags, _row_lo, _row_hi, edge_op);
        else
            hipLaunchKernelGGL((vgl_k_advance_static<false, E>), dim3(tiles), dim3(VGL_BLOCK), 0, stream, vertex_pointers, adjacent_ids, tile_row, edges_count, _process_shift,
                               frontier_flags, _row_lo, _row_hi, edge_op);
    }

    // the same, one lane per active vertex (safe stores): pre, the vertex's edges in adjacency order, post
    template <typename GraphContainer, typename FrontierContainer, typename EdgeOp, typename PreOp, typename PostOp>
    void rows_pass(GraphContainer &_graph, FrontierContainer &_frontier, long long _process_shift, int _row_lo, int _row_hi, EdgeOp &&edge_op, PreOp &&pre_op, PostOp &&post_op)
    {
        using E = typename std::decay<EdgeOp>::type; using P = typename std::decay<PreOp>::type; using Q = typename std::decay<PostOp>::type;
        LOAD_FRONTIER_DATA(_frontier);
        const graph_binding &gb = binding_of(_graph);
        const int vertices_count = _graph.get_vertices_count();
        const FrontierSparsityType t = _frontier.get_sparsity_type();
        if (_row_hi <= _row_lo) return;
        if (t == ALL_ACTIVE_FRONTIER)
            hipLaunchKernelGGL((vgl_k_advance_rows<0, E, P, Q>), dim3(grid_for(vertices_count)), dim3(VGL_BLOCK), 0, stream, vertices_count, gb.d_vertex_pointers, gb.d_adjacent_ids,
                               frontier_flags, frontier_ids, _process_shift, _row_lo, _row_hi, edge_op, pre_op, post_op);
        else if (t == DENSE_FRONTIER)
            hipLaunchKernelGGL((vgl_k_advance_rows<1, E, P, Q>), dim3(grid_for(vertices_count)), dim3(VGL_BLOCK), 0, stream, vertices_count, gb.d_vertex_pointers, gb.d_adjacent_ids,
                               frontier_flags, frontier_ids, _process_shift, _row_lo, _row_hi, edge_op, pre_op, post_op);
        else if (frontier_size > 0)
            hipLaunchKernelGGL((vgl_k_advance_rows<2, E, P, Q>), dim3(grid_for(frontier_size)), dim3(VGL_BLOCK), 0, stream, frontier_size, gb.d_vertex_pointers, gb.d_adjacent_ids,
                               frontier_flags, frontier_ids, _process_shift, _row_lo, _row_hi, edge_op, pre_op, post_op);
    }

    // compute inner implementation
    template <typename ComputeOperation, typename GraphContainer, typename FrontierContainer>
    void compute_worker(GraphContainer &_graph, FrontierContainer &_frontier, ComputeOperation &&compute_op);
    template <typename ComputeOperation>
    void compute_worker(CSRGraph &_graph, FrontierCSR &_frontier, ComputeOperation &&compute_op)
    { compute_on_csr_pointers(_graph, _frontier, compute_op); }
    template <typename ComputeOperation>
    void compute_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, ComputeOperation &&compute_op)
    { compute_on_csr_pointers(_graph, _frontier, compute_op); }
    template <typename ComputeOperation>
    void compute_worker(EdgesListGraph &_graph, FrontierEdgesList &_frontier, ComputeOperation &&compute_op)
    {
        LOAD_FRONTIER_DATA(_frontier);
        hip_shadows_to_device(stream);
        const int vertices_count = _graph.get_vertices_count();
        vertex_pass(vertices_count, edges_list_of(_graph).d_degree_prefix, _frontier.get_sparsity_type(), frontier_flags, frontier_ids, frontier_size, 0, vertices_count, compute_op);
        finish();
    }
    template <typename ComputeOperation, typename GraphContainer, typename FrontierContainer>
    void compute_on_csr_pointers(GraphContainer &_graph, FrontierContainer &_frontier, ComputeOperation &&compute_op)
    {
        LOAD_FRONTIER_DATA(_frontier);
        hip_shadows_to_device(stream);
        const int vertices_count = _graph.get_vertices_count();
        vertex_pass(vertices_count, binding_of(_graph).d_vertex_pointers, _frontier.get_sparsity_type(), frontier_flags, frontier_ids, frontier_size, 0, vertices_count, compute_op);
        finish();
    }

    // reduce inner implementation
    template <typename _T, typename ReduceOperation, typename GraphContainer, typename FrontierContainer>
    void reduce_worker(GraphContainer &_graph, FrontierContainer &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result);
    template <typename _T, typename ReduceOperation>
    void reduce_worker(CSRGraph &_graph, FrontierCSR &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
    { reduce_on_csr_pointers(_graph, _frontier, reduce_op, _reduce_type, _result); }
    template <typename _T, typename ReduceOperation>
    void reduce_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
    { reduce_on_csr_pointers(_graph, _frontier, reduce_op, _reduce_type, _result); }
    template <typename _T, typename ReduceOperation>
    void reduce_worker(EdgesListGraph &_graph, FrontierEdgesList &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
    { reduce_on_pointers(edges_list_of(_graph).d_degree_prefix, _graph, _frontier, reduce_op, _reduce_type, _result); }
    template <typename _T, typename ReduceOperation, typename GraphContainer, typename FrontierContainer>
    void reduce_on_csr_pointers(GraphContainer &_graph, FrontierContainer &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
    { reduce_on_pointers(binding_of(_graph).d_vertex_pointers, _graph, _frontier, reduce_op, _reduce_type, _result); }
    template <typename _T, typename ReduceOperation, typename GraphContainer, typename FrontierContainer>
    void reduce_on_pointers(const long long *vertex_pointers, GraphContainer &_graph, FrontierContainer &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
    {
        if (_reduce_type != REDUCE_SUM && _reduce_type != REDUCE_MAX) throw "Error in GraphAbstractionsHIP::reduce_worker: unsupported reduce type";   // multicore/reduce.hpp:144-150
        using R = typename std::decay<ReduceOperation>::type;
        LOAD_FRONTIER_DATA(_frontier);
        hip_shadows_to_device(stream);
        const FrontierSparsityType t = _frontier.get_sparsity_type();
        const int n = t == SPARSE_FRONTIER ? frontier_size : _graph.get_vertices_count();
        _result = 0;
        if (n <= 0) return;
        const int nb = (int)std::min<long long>(1024, ((long long)n + VGL_BLOCK - 1) / VGL_BLOCK);
        const bool mx = _reduce_type == REDUCE_MAX;
#define VGL_BIND_REDUCE(MODE)                                                                                                                                     \
        do {                                                                                                                                                      \
            if (mx) hipLaunchKernelGGL((vgl_k_reduce_partials<MODE, true, R>), dim3(nb), dim3(VGL_BLOCK), 0, stream, n, vertex_pointers, frontier_flags, frontier_ids, reduce_op, reduce_partials); \
            else hipLaunchKernelGGL((vgl_k_reduce_partials<MODE, false, R>), dim3(nb), dim3(VGL_BLOCK), 0, stream, n, vertex_pointers, frontier_flags, frontier_ids, reduce_op, reduce_partials); \
        } while (0)
        if (t == ALL_ACTIVE_FRONTIER) VGL_BIND_REDUCE(0);
        else if (t == DENSE_FRONTIER) VGL_BIND_REDUCE(1);
        else VGL_BIND_REDUCE(2);
#undef VGL_BIND_REDUCE
        VGL_HIP_BIND_RT(hipGetLastError());
        double r = 0.0;
        if (mx) {
            hipLaunchKernelGGL(vgl_k_max_fold, dim3(1), dim3(VGL_BLOCK), 0, stream, nb, (const double *)reduce_partials, reduce_partials + 1024);
            VGL_HIP_BIND_RT(hipGetLastError());
            VGL_HIP_BIND_CALL(vgl_hip_memcpy_d2h(ctx, &r, reduce_partials + 1024, sizeof(double)));
        } else
            VGL_HIP_BIND_CALL(vgl_hip_reduce_sum_f64_buffer(ctx, nb, reduce_partials, &r));      // fixed-order fold of the partials
        _result = (_T)r;
    }

    // advance inner implementation
    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void advance_worker(CSRGraph &_graph, FrontierCSR &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                        VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                        CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                        CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing);

    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void advance_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                        VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                        CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                        CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing);

    // EDGES_LIST_GRAPH: the operator over every stored edge; pre / post / collective operators and the frontier are not consulted
    // (multicore/advance_worker.hpp:10-57, gpu/advance.hpp:44-68)
    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void advance_worker(EdgesListGraph &_graph, FrontierEdgesList &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                        VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                        CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                        CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing)
    {
        using E = typename std::decay<EdgeOperation>::type;
        Timer tm;
        tm.start();
        hip_shadows_to_device(stream);
        const edges_list_binding &eb = edges_list_of(_graph);
        const long long edges_count = _graph.get_edges_count();
        if (edges_count > 0)
            hipLaunchKernelGGL((vgl_k_advance_edges_list<E>), dim3(grid_for(edges_count)), dim3(VGL_BLOCK), 0, stream, edges_count, _graph.get_vertices_count(),
                               (const int *)eb.d_src_ids, (const int *)eb.d_dst_ids, edge_op);
        finish();
        tm.end();
        performance_stats.update_advance_stats(tm.get_time(), edges_count * (INT_ELEMENTS_PER_EDGE + 1) * sizeof(int), edges_count);
    }

    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation, typename GraphContainer, typename FrontierContainer>
    void advance_worker(GraphContainer &_graph, FrontierContainer &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                        VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                        CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                        CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing)
    { throw "Error in GraphAbstractionsHIP::advance : this graph container is not served by the HIP backend (CSR_GRAPH and VECTOR_CSR_GRAPH are)"; }

public:
    // attaches graph-processing API to the specific graph
    GraphAbstractionsHIP(VGL_Graph &_graph, TraversalDirection _initial_traversal = SCATTER);
    ~GraphAbstractionsHIP();
    void enable_safe_stores() { safe_stores = true; }
    void disable_safe_stores() { safe_stores = false; }
    // the device copies of every container's adjacency are dropped (the next primitive copies again): for host code that rewrote a container's
    // vertex_pointers / adjacent_ids IN PLACE after a primitive had used it
    void forget_graph_bindings()
    {
        hipStreamSynchronize(stream);
        drop_frontier_handles();
        for (auto &kv : graph_bindings()) release(kv.second);
        graph_bindings().clear();
        frontier_generation = ++bindings_generation();
    }

    // generate new frontier implementation (public: it instantiates kernels on device lambdas, as in graph_abstractions_gpu.h:110-131)
    template <typename FilterCondition>
    void generate_new_frontier_worker(CSRGraph &_graph, FrontierCSR &_frontier, FilterCondition &&filter_cond);
    template <typename FilterCondition>
    void generate_new_frontier_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, FilterCondition &&filter_cond);
    // EDGES_LIST_GRAPH (multicore/generate_new_frontier.hpp:235-270): flags from the condition, size = how many passed, the frontier stays ALL_ACTIVE
    template <typename FilterCondition>
    void generate_new_frontier_worker(EdgesListGraph &_graph, FrontierEdgesList &_frontier, FilterCondition &&filter_cond)
    {
        using C = typename std::decay<FilterCondition>::type;
        Timer tm;
        tm.start();
        hip_shadows_to_device(stream);
        _frontier.set_direction(current_traversal_direction);
        const int vertices_count = _graph.get_vertices_count();
        unsigned long long passed = 0;
        VGL_HIP_BIND_RT(hipMemsetAsync(part_counters, 0, sizeof(unsigned long long), stream));
        if (vertices_count > 0)
            hipLaunchKernelGGL((vgl_k_edges_list_flags<C>), dim3(std::min(grid_for(vertices_count), 1024)), dim3(VGL_BLOCK), 0, stream, vertices_count, filter_cond, _frontier.flags, part_counters);
        VGL_HIP_BIND_RT(hipMemcpyAsync(&passed, part_counters, sizeof(passed), hipMemcpyDeviceToHost, stream));
        finish();
        _frontier.size = (int)passed;
        _frontier.neighbours_count = 0;
        _frontier.sparsity_type = ALL_ACTIVE_FRONTIER;
        tm.end();
        performance_stats.update_gnf_time(tm);
        performance_stats.update_bytes_requested((long long)vertices_count * 2.0 * sizeof(int));
    }
    template <typename FilterCondition, typename GraphContainer, typename FrontierContainer>
    void generate_new_frontier_worker(GraphContainer &_graph, FrontierContainer &_frontier, FilterCondition &&filter_cond)
    { throw "Error in GraphAbstractionsHIP::generate_new_frontier : this graph container is not served by the HIP backend"; }

    // performs user-defined "edge_op" operation over all OUTGOING edges, neighbouring specified frontier
    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void scatter(VGL_Graph &_graph, VGL_Frontier &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                 VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                 CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op, CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op)
    {
        this->common_scatter(_graph, _frontier, edge_op, vertex_preprocess_op, vertex_postprocess_op, collective_edge_op, collective_vertex_preprocess_op,
                             collective_vertex_postprocess_op, this);
    }
    template <typename EdgeOperation>
    void scatter(VGL_Graph &_graph, VGL_Frontier &_frontier, EdgeOperation &&edge_op)
    { scatter(_graph, _frontier, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP); }

    // performs user-defined "edge_op" operation over all INCOMING edges, neighbouring specified frontier
    template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
              typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
    void gather(VGL_Graph &_graph, VGL_Frontier &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op, CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op)
    {
        this->common_gather(_graph, _frontier, edge_op, vertex_preprocess_op, vertex_postprocess_op, collective_edge_op, collective_vertex_preprocess_op,
                            collective_vertex_postprocess_op, this);
    }
    template <typename EdgeOperation>
    void gather(VGL_Graph &_graph, VGL_Frontier &_frontier, EdgeOperation &&edge_op)
    { gather(_graph, _frontier, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP, edge_op, EMPTY_VERTEX_OP, EMPTY_VERTEX_OP); }

    // performs user-defined "compute_op" operation for each element in the given frontier
    template <typename ComputeOperation>
    void compute(VGL_Graph &_graph, VGL_Frontier &_frontier, ComputeOperation &&compute_op) { this->common_compute(_graph, _frontier, compute_op, this); }

    // performs reduction using user-defined "reduce_op" operation for each element in the given frontier
    template <typename _T, typename ReduceOperation>
    _T reduce(VGL_Graph &_graph, VGL_Frontier &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type)
    {
        _T result = 0;
        this->common_reduce(_graph, _frontier, reduce_op, _reduce_type, result, this);
        return result;
    }

    // creates new frontier, which satisfy user-defined "cond" condition
    template <typename FilterCondition>
    void generate_new_frontier(VGL_Graph &_graph, VGL_Frontier &_frontier, FilterCondition &&filter_cond)
    { this->common_generate_new_frontier(_graph, _frontier, filter_cond, this); }

    friend class GraphAbstractions;
};

// one library context per process (the reference's runtime has no handle to keep it in)
inline vgl_hip_ctx *vgl_hip_binding_context()
{
    static vgl_hip_ctx *c = nullptr;
    if (!c) VGL_HIP_BIND_CALL(vgl_hip_ctx_create(0, nullptr, &c));
    return c;
}

GraphAbstractionsHIP::GraphAbstractionsHIP(VGL_Graph &_graph, TraversalDirection _initial_traversal)
{
    processed_graph_ptr = &_graph;
    current_traversal_direction = _initial_traversal;
    ctx = vgl_hip_binding_context();
    stream = (hipStream_t)vgl_hip_ctx_stream(ctx);
    VGL_HIP_BIND_RT(hipMalloc((void **)&reduce_partials, sizeof(double) * (1024 + 8)));
    VGL_HIP_BIND_RT(hipMalloc((void **)&part_counters, sizeof(unsigned long long) * 8));
    VGL_HIP_BIND_RT(hipHostMalloc((void **)&part_sizes, sizeof(unsigned long long) * 8, hipHostMallocDefault));
    frontier_generation = bindings_generation();
    // the device copies of both direction containers are made HERE, before the algorithm starts its timer (the reference's algorithms construct the class
    // first and call _graph.move_to_device() next, bfs.hpp:58-72): the first primitive does not pay for 2 x (adjacency + vector extension) over PCIe
    if (_graph.get_container_type() == CSR_GRAPH) {
        binding_of(*(CSRGraph *)_graph.get_outgoing_data());
        if (_graph.get_number_of_directions() == BOTH_DIRECTIONS) binding_of(*(CSRGraph *)_graph.get_incoming_data());
        VGL_HIP_BIND_RT(hipStreamSynchronize(stream));
    } else if (_graph.get_container_type() == VECTOR_CSR_GRAPH) {
        binding_of(*(VectorCSRGraph *)_graph.get_outgoing_data());
        if (_graph.get_number_of_directions() == BOTH_DIRECTIONS) binding_of(*(VectorCSRGraph *)_graph.get_incoming_data());
        VGL_HIP_BIND_RT(hipStreamSynchronize(stream));
    }
}

GraphAbstractionsHIP::~GraphAbstractionsHIP()
{
    hipStreamSynchronize(stream);
    drop_frontier_handles();
    // (the graph bindings stay: see graph_bindings())
    hipFree(reduce_partials);
    hipFree(part_counters);
    hipHostFree(part_sizes);
}

template <typename ComputeOperation, typename GraphContainer, typename FrontierContainer>
void GraphAbstractionsHIP::compute_worker(GraphContainer &_graph, FrontierContainer &_frontier, ComputeOperation &&compute_op)
{ throw "Error in GraphAbstractionsHIP::compute : this graph container is not served by the HIP backend"; }

template <typename _T, typename ReduceOperation, typename GraphContainer, typename FrontierContainer>
void GraphAbstractionsHIP::reduce_worker(GraphContainer &_graph, FrontierContainer &_frontier, ReduceOperation &&reduce_op, REDUCE_TYPE _reduce_type, _T &_result)
{ throw "Error in GraphAbstractionsHIP::reduce : this graph container is not served by the HIP backend"; }

// CSR_GRAPH: pre -> every edge of the active vertices -> post; the collective set is never called (advance_worker.hpp:62-149)
template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
          typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
void GraphAbstractionsHIP::advance_worker(CSRGraph &_graph, FrontierCSR &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                                          VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                                          CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                                          CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing)
{
    Timer tm;
    tm.start();
    hip_shadows_to_device(stream);
    const int vertices_count = _graph.get_vertices_count();
    const long long edges_count = _graph.get_edges_count();
    const long long *vertex_pointers = binding_of(_graph).d_vertex_pointers;       // the device copy (the kernels' view of the container)
    LOAD_FRONTIER_DATA(_frontier);
    const long long process_shift = compute_process_shift(current_traversal_direction, CSR_STORAGE);
    const FrontierSparsityType t = _frontier.get_sparsity_type();
    if (safe_stores) rows_pass(_graph, _frontier, process_shift, 0, vertices_count, edge_op, vertex_preprocess_op, vertex_postprocess_op);
    else {
        if (!is_empty_op<VertexPreprocessOperation>()) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, vertices_count, vertex_preprocess_op);
        edge_pass(_graph, _frontier, process_shift, 0, vertices_count, edge_op);
        if (!is_empty_op<VertexPostprocessOperation>()) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, vertices_count, vertex_postprocess_op);
    }
    finish();
    tm.end();
    const long long work = t == ALL_ACTIVE_FRONTIER ? edges_count : frontier_neighbours_count;
    performance_stats.update_advance_stats(tm.get_time(), work * INT_ELEMENTS_PER_EDGE * sizeof(int), work);
}

// VECTOR_CSR_GRAPH (advance_worker.hpp:204-319)
template <typename EdgeOperation, typename VertexPreprocessOperation, typename VertexPostprocessOperation, typename CollectiveEdgeOperation,
          typename CollectiveVertexPreprocessOperation, typename CollectiveVertexPostprocessOperation>
void GraphAbstractionsHIP::advance_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, EdgeOperation &&edge_op, VertexPreprocessOperation &&vertex_preprocess_op,
                                          VertexPostprocessOperation &&vertex_postprocess_op, CollectiveEdgeOperation &&collective_edge_op,
                                          CollectiveVertexPreprocessOperation &&collective_vertex_preprocess_op,
                                          CollectiveVertexPostprocessOperation &&collective_vertex_postprocess_op, bool _inner_mpi_processing)
{
    static_assert(VECTOR_LENGTH == 64, "the vector-extension kernel maps one segment to one 64-lane wavefront");
    using E = typename std::decay<EdgeOperation>::type; using CE = typename std::decay<CollectiveEdgeOperation>::type;
    using P = typename std::decay<VertexPreprocessOperation>::type; using CP = typename std::decay<CollectiveVertexPreprocessOperation>::type;
    using Q = typename std::decay<VertexPostprocessOperation>::type; using CQ = typename std::decay<CollectiveVertexPostprocessOperation>::type;
    Timer tm;
    tm.start();
    hip_shadows_to_device(stream);
    const graph_binding &gb = binding_of(_graph);
    const int vertices_count = _graph.get_vertices_count();
    const long long edges_count = _graph.get_edges_count();
    const long long *vertex_pointers = gb.d_vertex_pointers;                        // device copies: CSR part and vector extension
    const long long *ve_vector_group_ptrs = gb.d_ve_group_ptrs;
    const int *ve_vector_group_sizes = gb.d_ve_group_sizes, *ve_adjacent_ids = gb.d_ve_adjacent_ids;
    const int ve_starting_vertex = _graph.get_ve_ptr()->get_starting_vertex(), ve_vector_segments_count = _graph.get_ve_ptr()->get_vector_segments_count();
    LOAD_FRONTIER_DATA(_frontier);
    const int collective_start = _graph.get_vector_core_threshold_vertex();       // [0, collective_start): vector engine + vector core ranges
    const long long csr_shift = compute_process_shift(current_traversal_direction, CSR_STORAGE);
    const long long ve_shift = compute_process_shift(current_traversal_direction, VE_STORAGE);
    const FrontierSparsityType t = _frontier.get_sparsity_type();
    long long work = 0;
    if (t == SPARSE_FRONTIER) {
        // every part of a sparse frontier reads the CSR storage (advance_sparse.hpp:24,90,150)
        const vgl_range_vertex_op<P, CP> pre{vertex_preprocess_op, collective_vertex_preprocess_op, collective_start};
        const vgl_range_vertex_op<Q, CQ> post{vertex_postprocess_op, collective_vertex_postprocess_op, collective_start};
        const vgl_range_edge_op<E, CE> both{edge_op, collective_edge_op, collective_start};
        const bool no_pre = is_empty_op<VertexPreprocessOperation>() && is_empty_op<CollectiveVertexPreprocessOperation>();
        const bool no_post = is_empty_op<VertexPostprocessOperation>() && is_empty_op<CollectiveVertexPostprocessOperation>();
        if (safe_stores) rows_pass(_graph, _frontier, csr_shift, 0, vertices_count, both, pre, post);
        else {
            if (!no_pre) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, vertices_count, pre);
            edge_pass(_graph, _frontier, csr_shift, 0, vertices_count, both);
            if (!no_post) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, vertices_count, post);
        }
        work = frontier_neighbours_count;
    } else {
        if (collective_start > 0 && safe_stores) rows_pass(_graph, _frontier, csr_shift, 0, collective_start, edge_op, vertex_preprocess_op, vertex_postprocess_op);
        else if (collective_start > 0) {
            if (!is_empty_op<VertexPreprocessOperation>()) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, collective_start, vertex_preprocess_op);
            edge_pass(_graph, _frontier, csr_shift, 0, collective_start, edge_op);
            if (!is_empty_op<VertexPostprocessOperation>()) vertex_pass(vertices_count, vertex_pointers, t, frontier_flags, frontier_ids, frontier_size, 0, collective_start, vertex_postprocess_op);
        }
        if (ve_vector_segments_count > 0) {
            const unsigned blocks = (unsigned)(((long long)ve_vector_segments_count * 64 + VGL_BLOCK - 1) / VGL_BLOCK);
            if (t == DENSE_FRONTIER)
                hipLaunchKernelGGL((vgl_k_advance_vector_extension<true, CE, CP, CQ>), dim3(blocks), dim3(VGL_BLOCK), 0, stream, ve_vector_segments_count, ve_starting_vertex, vertices_count,
                                   vertex_pointers, ve_vector_group_ptrs, ve_vector_group_sizes, ve_adjacent_ids, frontier_flags, ve_shift, collective_edge_op,
                                   collective_vertex_preprocess_op, collective_vertex_postprocess_op);
            else
                hipLaunchKernelGGL((vgl_k_advance_vector_extension<false, CE, CP, CQ>), dim3(blocks), dim3(VGL_BLOCK), 0, stream, ve_vector_segments_count, ve_starting_vertex, vertices_count,
                                   vertex_pointers, ve_vector_group_ptrs, ve_vector_group_sizes, ve_adjacent_ids, frontier_flags, ve_shift, collective_edge_op,
                                   collective_vertex_preprocess_op, collective_vertex_postprocess_op);
        }
        work = t == ALL_ACTIVE_FRONTIER ? edges_count : frontier_neighbours_count;
    }
    finish();
    tm.end();
    performance_stats.update_advance_stats(tm.get_time(), work * INT_ELEMENTS_PER_EDGE * sizeof(int), work);
}

// CSR_GRAPH frontiers: ALL_ACTIVE when every vertex passes, else SPARSE with ascending ids (generate_new_frontier.hpp:113-164)
template <typename FilterCondition>
void GraphAbstractionsHIP::generate_new_frontier_worker(CSRGraph &_graph, FrontierCSR &_frontier, FilterCondition &&filter_cond)
{
    using C = typename std::decay<FilterCondition>::type;
    Timer tm;
    tm.start();
    hip_shadows_to_device(stream);
    _frontier.set_direction(current_traversal_direction);
    const int vertices_count = _graph.get_vertices_count();
    vgl_hip_graph *gh = handle_of(_graph);
    vgl_hip_frontier *fh = handle_of(_frontier, gh);
    vgl_hip_gnf_buffers b;
    VGL_HIP_BIND_CALL(vgl_hip_gnf_begin(ctx, gh, fh, 1, &b));
    const vgl_pred_user<C> pred{filter_cond, binding_of(_graph).d_vertex_pointers};
    hipLaunchKernelGGL((vgl_k_gnf_count<vgl_pred_user<C>>), dim3((unsigned)b.nvtiles), dim3(VGL_BLOCK), 0, stream, pred, b.nrows, b.row_begin, b.out_rowptr, b.vt_cnt, b.vt_deg,
                       b.front_bytes, (uint8_t *)nullptr, b.flags, b.ticket, b.vt_cnt_off, b.vt_deg_off, b.counters, b.plan_offs, b.host_counters, b.seq);
    VGL_HIP_BIND_RT(hipGetLastError());
    VGL_HIP_BIND_CALL(vgl_hip_gnf_complete(ctx, gh, fh, 0.0, 1, b.seq));
    int32_t size = 0; int64_t neighbours = 0; int sparsity = 0;
    VGL_HIP_BIND_CALL(vgl_hip_frontier_info(ctx, fh, &size, &neighbours, &sparsity));
    finish();
    _frontier.size = size;
    _frontier.neighbours_count = neighbours;
    _frontier.sparsity_type = sparsity == VGL_HIP_FRONTIER_ALL_ACTIVE ? ALL_ACTIVE_FRONTIER : SPARSE_FRONTIER;
    stamp_generated(_frontier, gh, sparsity);
    tm.end();
    performance_stats.update_gnf_time(tm);
    performance_stats.update_bytes_requested((long long)vertices_count * 4.0 * sizeof(int));
}

// VECTOR_CSR_GRAPH frontiers: ALL_ACTIVE, DENSE (flags only) above 0.7 of the vertices, else SPARSE; sizes and degree sums per degree range
// (generate_new_frontier.hpp:29-111)
template <typename FilterCondition>
void GraphAbstractionsHIP::generate_new_frontier_worker(VectorCSRGraph &_graph, FrontierVectorCSR &_frontier, FilterCondition &&filter_cond)
{
    using C = typename std::decay<FilterCondition>::type;
    Timer tm;
    tm.start();
    hip_shadows_to_device(stream);
    const int vertices_count = _graph.get_vertices_count();
    vgl_hip_graph *gh = handle_of(_graph);
    vgl_hip_frontier *fh = handle_of(_frontier, gh);
    vgl_hip_gnf_buffers b;
    VGL_HIP_BIND_CALL(vgl_hip_gnf_begin(ctx, gh, fh, 1, &b));
    const vgl_pred_user<C> pred{filter_cond, binding_of(_graph).d_vertex_pointers};
    hipLaunchKernelGGL((vgl_k_gnf_count<vgl_pred_user<C>>), dim3((unsigned)b.nvtiles), dim3(VGL_BLOCK), 0, stream, pred, b.nrows, b.row_begin, b.out_rowptr, b.vt_cnt, b.vt_deg,
                       b.front_bytes, (uint8_t *)nullptr, b.flags, b.ticket, b.vt_cnt_off, b.vt_deg_off, b.counters, b.plan_offs, b.host_counters, b.seq);
    VGL_HIP_BIND_RT(hipGetLastError());
    // the three parts' sizes from the per-tile counts the pass above leaves (stream order: after it, before the host looks), while the host waits for the totals
    hipLaunchKernelGGL(vgl_k_frontier_parts_from_tiles, dim3(1), dim3(VGL_PARTS_THREADS), 0, stream, vertices_count, (long long)b.nvtiles, (const int *)b.vt_cnt,
                       (const long long *)b.vt_deg, (const unsigned char *)b.front_bytes, (const long long *)binding_of(_graph).d_vertex_pointers,
                       _graph.get_vector_engine_threshold_vertex(), _graph.get_vector_core_threshold_vertex(), part_sizes);
    VGL_HIP_BIND_RT(hipGetLastError());
    VGL_HIP_BIND_CALL(vgl_hip_gnf_complete(ctx, gh, fh, 0.7, 1, b.seq));
    int32_t size = 0; int64_t neighbours = 0; int sparsity = 0;
    VGL_HIP_BIND_CALL(vgl_hip_frontier_info(ctx, fh, &size, &neighbours, &sparsity));
    finish();                                       // (the kernel's stores to the pinned words are visible once the stream has drained)
    const volatile unsigned long long *parts = part_sizes;
    _frontier.vector_engine_part_size = (int)parts[0]; _frontier.vector_engine_part_neighbours_count = (long long)parts[3];
    _frontier.vector_core_part_size = (int)parts[1]; _frontier.vector_core_part_neighbours_count = (long long)parts[4];
    _frontier.collective_part_size = (int)parts[2]; _frontier.collective_part_neighbours_count = (long long)parts[5];
    _frontier.size = size;
    _frontier.neighbours_count = neighbours;
    if (sparsity == VGL_HIP_FRONTIER_ALL_ACTIVE) _frontier.sparsity_type = ALL_ACTIVE_FRONTIER;
    else {
        const FrontierSparsityType t = sparsity == VGL_HIP_FRONTIER_DENSE ? DENSE_FRONTIER : SPARSE_FRONTIER;
        _frontier.sparsity_type = t; _frontier.vector_engine_part_type = t; _frontier.vector_core_part_type = t; _frontier.collective_part_type = t;
    }
    stamp_generated(_frontier, gh, sparsity);
    tm.end();
    performance_stats.update_gnf_time(tm);
    performance_stats.update_bytes_requested((long long)vertices_count * 2.0 * sizeof(int));
}
