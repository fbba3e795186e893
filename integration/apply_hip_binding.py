#!/usr/bin/env python3
"""Adds the __USE_HIP__ architecture to a COPY of the VGL source tree.

    python3 integration/apply_hip_binding.py /tmp/vgl_with_hip        # the directory holds a copy of the reference checkout

What a maintainer would commit, expressed as anchored edits instead of a unified diff (a diff would carry lines of the reference's source as
context; the anchors below are the shortest strings that locate each edit):
  * the new backend directory vgl_compute_api/hip/ (copied from integration/vgl_compute_api/hip/)
  * architecture_independent_api.h : device-lambda argument macros, VGL_GRAPH_ABSTRACTIONS, the atomic VGL_SRC_ID_ADD / VGL_INC / VGL_DEC
  * settings.h                      : VECTOR_LENGTH = 64 (one wavefront), thresholds
  * common dispatch (advance.hpp, compute.hpp) : call the workers directly, not inside an OpenMP parallel region
  * common/graph_abstractions.h     : include the backend header
  * MemoryAPI                       : hipMallocManaged / hipFree for containers; user arrays are SHADOWED (pinned host mirror + HBM buffer + owner flag,
                                      vgl_compute_api/hip/shadow_memory.h); frontier flags / ids in device memory
  * VerticesArray / EdgesArray      : device code indexes the HBM buffer, host code the mirror (fetched back on demand)
  * frontier containers             : friend class GraphAbstractionsHIP; the plan stamp (hip_plan_token) and its reset in every mutator
  * move_to_device / move_to_host   : the CUDA flavour's family exists; on user arrays they are real copies mirror <-> HBM
  * algorithms/{pr,sssp,cc}         : the reference's GPU variants (gpu_pr.hpp, gpu_shortest_paths.hpp, gpu_shiloach_vishkin.hpp) compile for __USE_HIP__;
                                      three CUDA runtime calls by name get a HIP branch
  * algorithms/coloring             : bit helpers callable from device code
  * algorithms/tc                   : ParallelPrimitives::copy_if_indexes evaluates a device condition in kernels; EDGES_LIST_GRAPH served by the class;
                                      the condensed graph gets the vertex its largest component id needs
(the hipcc command line is in oracle/Makefile, target `binding`)
Every rule must apply (an anchor that is not found is an error): the script is also the test that the reference still has the shape the
binding was written against.  tests/test_reference_binding.py applies it to a copy in /tmp and compiles seven of the reference's apps with hipcc."""
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GPU_OR_HIP = "#if defined(__USE_GPU__) || defined(__USE_HIP__)"

# (file, kind, anchor regex, text[, expected count[, which]])   kinds: sub = replace every match; after / before = insert a block next to the
# line of match number `which` (default 0)
RULES = [
    # ---- architecture selection ----
    ("architecture_independent_api.h", "sub", r"^#ifdef __USE_GPU__$(?!\n#define VGL_GRAPH_ABSTRACTIONS)", GPU_OR_HIP, 5),
    ("architecture_independent_api.h", "before", r"^#define VGL_FRONTIER VGL_Frontier",
     "#ifdef __USE_HIP__\n#define VGL_GRAPH_ABSTRACTIONS GraphAbstractionsHIP\n#endif\n\n"),
    ("settings.h", "before", r"^// ARM/Intel/AMD multicore properties",
     "// AMD Instinct (HIP) properties: a vector is one 64-lane wavefront\n"
     "/////////////////////////////////////////////////////////////////////////////////////////////////////////////////////\n\n"
     "#ifdef __USE_HIP__\n#define VECTOR_LENGTH 64\n#define VECTOR_LENGTH_POW 6\n#define MAX_SX_AURORA_THREADS 8\n#define LLC_CACHE_SIZE 4*1024*1024\n#endif\n\n"
     "/////////////////////////////////////////////////////////////////////////////////////////////////////////////////////\n"),
    # ---- common dispatch: workers launch kernels, they are not OpenMP bodies ----
    ("vgl_compute_api/common/advance.hpp", "sub", r"^(\s*)#ifdef __USE_GPU__$", r"\1" + GPU_OR_HIP, 8),
    ("vgl_compute_api/common/compute.hpp", "sub", r"^(\s*)#ifdef __USE_GPU__$", r"\1" + GPU_OR_HIP, 4),
    ("vgl_compute_api/common/graph_abstractions.h", "before", r"^#if defined\(__USE_MULTICORE__\)$",
     "#ifdef __USE_HIP__\n#include \"vgl_compute_api/hip/graph_abstractions_hip.h\"\n#endif\n\n"),
    # ---- coloring.hpp calls its bit helpers (clear_bit, smallest_bit_pos) from operator lambdas: they become callable from device code (clang, unlike
    #      nvcc, checks this in templates that are never instantiated too).  Its scatter under enable_safe_stores() -- a read-modify-write of per-vertex
    #      data without atomics -- runs one lane per vertex in the HIP class (GraphAbstractionsHIP::enable_safe_stores).
    #      ----
    ("algorithms/coloring/coloring.hpp", "sub", r"^inline (size_t|int) (set_bit|clear_bit|get_bit|smallest_bit_pos)\(",
     r"#if defined(__USE_GPU__) || defined(__USE_HIP__)\n__host__ __device__\n#endif\ninline \1 \2(", 4),
    # ---- tc.hpp (TransitiveClosure::vgl_purdoms) hands a DEVICE lambda to ParallelPrimitives::copy_if_indexes, whose only body is a host loop
    #      (copy_if/copy_if.hpp:285-298: the reference's own GPU flavour cannot build it): the condition runs in kernels (vgl_compute_api/hip/
    #      parallel_primitives_hip.h).  Its condensed graph is an EDGES_LIST_GRAPH: the class serves that container (graph_abstractions_hip.h), the
    #      container tells the backend when it frees its arrays.  One correction of the algorithm, for this architecture only: tc.hpp:104-110 sizes the
    #      condensed graph by the LARGEST component id (REDUCE_MAX) instead of the number of ids, so the component with that id is a vertex one past
    #      every array of the condensed graph -- an unnoticed stray access in a host build, not something a kernel may do ----
    ("vgl_runtime/helpers/parallel_primitives/primitives.h", "before", r'^#include "copy_if/copy_if.hpp"$',
     "#ifdef __USE_HIP__\n#include \"vgl_compute_api/hip/parallel_primitives_hip.h\"\n#endif\n"),
    ("vgl_runtime/helpers/parallel_primitives/copy_if/copy_if.hpp", "sub", r"^(\s*)#elif defined\(__USE_MULTICORE__\)\n(\s*num_elements = omp_copy_if_indexes\()",
     r"\1#elif defined(__USE_HIP__)\n\1num_elements = hip_copy_if_indexes(_cond, _out_data, _size, _index_offset);\n\1#elif defined(__USE_MULTICORE__)\n\2", 1),
    ("algorithms/tc/tc.hpp", "after", r"^\s*int new_vertices_count = graph_API\.reduce<int>\(_graph, frontier, max_component_num, REDUCE_MAX\);$",
     "    #ifdef __USE_HIP__\n    new_vertices_count += 1;          // ids are 0 .. max: the condensed graph has max + 1 vertices\n    #endif\n"),
    ("vgl_datastructures/graphs/undirected_containers/edges_list/edges_list_graph.hpp", "sub", r"^(void EdgesListGraph::free\(\)\n\{\n)",
     r"\1    #ifdef __USE_HIP__\n    hip_container_changed(this);\n    #endif\n", 1),
    # ---- memory: managed allocations, as the CUDA flavour (__USE_MANAGED_MEMORY__, settings.h) ----
    ("vgl_runtime/helpers/memory_API/memory_API.hpp", "before", r"^\s*#elif defined\(__USE_KNL__\)$",
     "    #elif defined(__USE_HIP__)\n    if(hipMallocManaged((void**)_ptr, _size * sizeof(_T)) != hipSuccess) throw \"Error in MemoryAPI::allocate_array : hipMallocManaged failed\";\n", 2, 0),
    ("vgl_runtime/helpers/memory_API/memory_API.hpp", "before", r"^\s*#elif defined\(__USE_KNL__\)$",
     "        #elif defined(__USE_HIP__)\n        hipFree((void*)_ptr);\n", 2, 1),
    ("vgl_runtime/helpers/memory_API/memory_API.h", "before", r"^class MemoryAPI", "#ifdef __USE_HIP__\n#include <hip/hip_runtime.h>\n#endif\n\n"),
    # ---- move_to_device / move_to_host of every container and user array: the CUDA flavour's declarations, definitions and call sites
    #      (algorithms/bfs/bfs.hpp:70-74, hits/hits.hpp:12-17, the gpu_*.hpp variants) exist under __USE_HIP__ too: for a user array they copy the
    #      pinned host mirror into its HBM buffer / back (shadow_memory.h), for managed graph containers and device-only frontier arrays they do nothing ----
    ("*", "sub_tree", r"#ifdef __USE_GPU__(?=\n(?:\s*template <typename _T>\n)?[^\n]*move_(?:array_)?to_(?:device|host))", GPU_OR_HIP, 43,
     ("vgl_runtime/helpers/memory_API/memory_API.hpp",)),
    ("vgl_runtime/helpers/memory_API/memory_API.hpp", "after", r"\A",
     "// HIP flavour: move_array_to_device / _to_host are the ownership transitions of a shadowed user array (vgl_compute_api/hip/shadow_memory.h) made early;\n"
     "// for anything else (graph containers in managed memory, frontier arrays in device memory) they are accepted and do nothing\n"
     "#ifdef __USE_HIP__\n"
     "template <typename _T>\nvoid MemoryAPI::move_array_to_device(_T *_ptr, size_t _size)\n{\n    if(HipShadow *s = hip_shadow_of(_ptr)) hip_shadow_acquire_device(s, 0);\n}\n\n"
     "template <typename _T>\nvoid MemoryAPI::move_array_to_host(_T *_ptr, size_t _size)\n{\n    if(HipShadow *s = hip_shadow_of(_ptr)) hip_shadow_host_access(s);\n}\n\n"
     "template <typename _T>\nvoid MemoryAPI::allocate_device_array(_T **_ptr, size_t _size)\n{\n"
     "    if(hipMalloc((void**)_ptr, (_size > 0 ? _size : 1) * sizeof(_T)) != hipSuccess) throw \"Error in MemoryAPI::allocate_device_array : hipMalloc failed\";\n}\n#endif\n\n"),
    # ---- the reference's own GPU variants of PageRank, SSSP and Shiloach-Vishkin (operators written for device lambdas; three CUDA runtime
    #      calls by name get a HIP branch) ----
    ("algorithms/pr/pr.h", "sub", r"^(\s*)#ifdef __USE_GPU__$", r"\1" + GPU_OR_HIP, 1),
    ("algorithms/pr/gpu_pr.hpp", "sub", r"^#ifdef __USE_GPU__$", GPU_OR_HIP, 1),
    ("algorithms/sssp/shortest_paths.h", "sub", r"^(\s*)#ifdef __USE_GPU__$", r"\1" + GPU_OR_HIP, 3),
    ("algorithms/sssp/gpu_shortest_paths.hpp", "sub", r"^#ifdef __USE_GPU__$", GPU_OR_HIP, 4),
    ("algorithms/cc/cc.h", "sub", r"^(\s*)#ifdef __USE_GPU__$", r"\1" + GPU_OR_HIP, 1),
    ("algorithms/cc/gpu_shiloach_vishkin.hpp", "sub", r"^#ifdef __USE_GPU__$", GPU_OR_HIP, 1),
    ("algorithms/sssp/gpu_shortest_paths.hpp", "sub", r"\bGraphAbstractionsGPU\b", "VGL_GRAPH_ABSTRACTIONS", 3),      # (the file names the CUDA class instead of the macro)
    ("algorithms/sssp/gpu_shortest_paths.hpp", "sub", r"^(\s*)cudaMemset\((.*)\);$",
     r"\1#ifdef __USE_HIP__\n\1(void)hipMemset(was_updated.get_device_ptr(), 0, sizeof(char) * _graph.get_vertices_count());\n\1#else\n\1cudaMemset(\2);\n\1#endif", 1),
    ("algorithms/cc/gpu_shiloach_vishkin.hpp", "sub", r"^(\s*)cudaMallocManaged\((.*)\);$",
     r"\1#ifdef __USE_HIP__\n\1(void)hipMallocManaged(\2);\n\1#else\n\1cudaMallocManaged(\2);\n\1#endif", 2),
    # ---- where the arrays live (vgl_compute_api/hip/shadow_memory.h).  Graph containers and one-word flags stay in managed (on this pool: host-resident)
    #      memory, which their host-side import and the sequential checkers read at full speed; the backend class keeps device copies of the adjacency
    #      it traverses.  VerticesArray / EdgesArray get a pinned host mirror AND a buffer in HBM with an owner flag: device lambdas index the HBM
    #      buffer, host code the mirror, a host accessor fetches the array back when kernels ran since, every primitive uploads what the host wrote
    #      since; move_to_device() / move_to_host() are those transitions made early.  Frontier flags / ids are device memory only ----
    ("vgl_runtime/helpers/memory_API/memory_API.h", "before", r"^class MemoryAPI",
     "#ifdef __USE_HIP__\n#include \"vgl_compute_api/hip/shadow_memory.h\"\n#define VGL_HOST_ACCESS(obj) hip_shadow_host_access((obj)->shadow)\n#else\n#define VGL_HOST_ACCESS(obj)\n#endif\n\n"),
    ("vgl_runtime/helpers/memory_API/memory_API.h", "after", r"static void resize\(_T \*\*_ptr, size_t _new_size\);",
     "\n    #ifdef __USE_HIP__\n    // arrays only kernels (and single host stores) touch: frontier flags / ids\n"
     "    template <typename _T>\n    static void allocate_device_array(_T **_ptr, size_t _size);\n    #endif\n"),
    # VerticesArray: members, accessors, constructors / destructor, the host-side member functions
    ("vgl_datastructures/vertices_array/vertices_array.h", "after", r"^\s*_T \*vertices_data;$",
     "    #ifdef __USE_HIP__\n    _T *device_data;          // the buffer in HBM (vertices_data is the pinned host mirror)\n    HipShadow *shadow;        // who holds the current values\n    #endif\n"),
    ("vgl_datastructures/vertices_array/vertices_array.h", "sub", r"^(\s*)#ifdef __USE_GPU__$(?=\n\s*__host__ __device__)",
     r"\1#if defined(__USE_HIP__)" "\n"
     r"\1// device code: the HBM buffer; host code: the mirror, fetched back first when kernels ran since" "\n"
     r"\1__host__ __device__ inline _T &at(int _idx) const" "\n" r"\1{" "\n"
     r"\1    #if defined(__HIP_DEVICE_COMPILE__)" "\n" r"\1    return device_data[_idx];" "\n"
     r"\1    #else" "\n" r"\1    hip_shadow_host_access(shadow); return vertices_data[_idx];" "\n" r"\1    #endif" "\n" r"\1}" "\n"
     r"\1__host__ __device__ inline _T get(int _idx) const { return at(_idx); };" "\n"
     r"\1__host__ __device__ inline void set(int _idx, _T _val) const { at(_idx) = _val; };" "\n"
     r"\1__host__ __device__ inline _T& operator[] (int _idx) const { return at(_idx); };" "\n"
     r"\1#elif defined(__USE_GPU__)"),
    ("vgl_datastructures/vertices_array/vertices_array.h", "sub", r"^(\s*)(_T \*get_ptr\(\) \{return vertices_data;\};)$",
     r"\1#ifdef __USE_HIP__" "\n"
     r"\1_T *get_ptr() { hip_shadow_host_access(shadow); return vertices_data; };                 // host code: the mirror, current" "\n"
     r"\1_T *get_device_ptr() { hip_shadow_acquire_device(shadow, 0); return device_data; };      // runtime calls on the device buffer (hipMemset)" "\n"
     r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/vertices_array/vertices_array.hpp", "sub", r"^(\s*)(MemoryAPI::allocate_array\(&this->vertices_data, this->vertices_count\);)$",
     r"\1#ifdef __USE_HIP__" "\n"
     r"\1shadow = hip_shadow_allocate(sizeof(_T) * (size_t)this->vertices_count);" "\n"
     r"\1this->vertices_data = (_T*)shadow->host; this->device_data = (_T*)shadow->device;" "\n"
     r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/vertices_array/vertices_array.hpp", "after", r"^\s*this->vertices_data = _copy_obj\.vertices_data;$",
     "    #ifdef __USE_HIP__\n    this->device_data = _copy_obj.device_data;\n    this->shadow = _copy_obj.shadow;\n    #endif\n"),
    ("vgl_datastructures/vertices_array/vertices_array.hpp", "sub", r"^(\s*)(MemoryAPI::free_array\(this->vertices_data\);)$",
     r"\1#ifdef __USE_HIP__" "\n" r"\1hip_shadow_free(shadow); shadow = NULL; device_data = NULL;" "\n" r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/vertices_array/vertices_array.hpp", "sub", r"^(void VerticesArray<_T>::(?:set_all_random|print)\([^)]*\)\n\{\n)",
     r"\1    VGL_HOST_ACCESS(this);\n", 3),
    # set_all_constant between primitives (coloring.hpp:90,113 does it every iteration): filled where the array lives
    ("vgl_datastructures/vertices_array/vertices_array.hpp", "sub", r"^(void VerticesArray<_T>::set_all_constant\([^)]*\)\n\{\n)(\s*)(MemoryAPI::set\(this->vertices_data, _const, this->vertices_count\);)$",
     r"\1\2#ifdef __USE_HIP__\n\2if(hip_shadow_fill(shadow, this->device_data, _const, (size_t)this->vertices_count)) return;\n\2VGL_HOST_ACCESS(this);\n\2#endif\n\2\3", 1),
    # EdgesArray: the same (its container classes keep pointers into the mirror: attach_pointer)
    ("vgl_datastructures/edges_array/edges_array.h", "after", r"^\s*_T \*edges_data;$",
     "    #ifdef __USE_HIP__\n    _T *device_data;          // the buffer in HBM (edges_data is the pinned host mirror)\n    HipShadow *shadow;\n    #endif\n"),
    ("vgl_datastructures/edges_array/edges_array.h", "sub", r"^(\s*)#ifdef __USE_GPU__$(?=\n\s*__host__ __device__)",
     r"\1#if defined(__USE_HIP__)" "\n"
     r"\1__host__ __device__ inline _T &at(long long _global_idx) const" "\n" r"\1{" "\n"
     r"\1    #if defined(__HIP_DEVICE_COMPILE__)" "\n" r"\1    return device_data[_global_idx];" "\n"
     r"\1    #else" "\n" r"\1    hip_shadow_host_access(shadow); return edges_data[_global_idx];" "\n" r"\1    #endif" "\n" r"\1}" "\n"
     r"\1__host__ __device__ inline _T get(long long _global_idx) const { return at(_global_idx); };" "\n"
     r"\1__host__ __device__ inline void set(long long _global_idx, _T _val) const { at(_global_idx) = _val; };" "\n"
     r"\1__host__ __device__ inline _T& operator[] (long long _global_idx) const { return at(_global_idx); };" "\n"
     r"\1#elif defined(__USE_GPU__)"),
    ("vgl_datastructures/edges_array/edges_array.h", "sub", r"^(\s*)(inline _T \*get_ptr\(\) const \{ return edges_data; \};)$",
     r"\1#ifdef __USE_HIP__" "\n"
     r"\1inline _T *get_ptr() const { hip_shadow_host_access(shadow); return edges_data; };" "\n"
     r"\1inline _T *get_device_ptr() const { hip_shadow_acquire_device(shadow, 0); return device_data; };" "\n"
     r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/edges_array/edges_array.h", "sub", r"\{ container->(set_all_constant|set_all_random|print)\(", r"{ VGL_HOST_ACCESS(this); container->\1(", 3),
    ("vgl_datastructures/edges_array/edges_array.hpp", "sub", r"^(\s*)(MemoryAPI::allocate_array\(&edges_data, container->get_total_array_size\(\)\);)$",
     r"\1#ifdef __USE_HIP__" "\n"
     r"\1shadow = hip_shadow_allocate(sizeof(_T) * (size_t)container->get_total_array_size());" "\n"
     r"\1edges_data = (_T*)shadow->host; device_data = (_T*)shadow->device;" "\n"
     r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/edges_array/edges_array.hpp", "after", r"^\s*this->edges_data = _copy_obj\.edges_data;$",
     "    #ifdef __USE_HIP__\n    this->device_data = _copy_obj.device_data;\n    this->shadow = _copy_obj.shadow;\n    #endif\n"),
    ("vgl_datastructures/edges_array/edges_array.hpp", "sub", r"^(\s*)(MemoryAPI::free_array\(edges_data\);)$",
     r"\1#ifdef __USE_HIP__" "\n" r"\1hip_shadow_free(shadow); shadow = NULL; device_data = NULL;" "\n" r"\1#else" "\n" r"\1\2" "\n" r"\1#endif"),
    ("vgl_datastructures/edges_array/edges_array.hpp", "sub", r"^(void EdgesArray<_T>::finalize_advance\([^)]*\)\n\{\n)", r"\1    VGL_HOST_ACCESS(this);\n", 1),
    # frontier flags / ids: device memory
    ("vgl_datastructures/frontier/containers/csr/frontier_csr.hpp", "sub", r"^(\s*)MemoryAPI::allocate_array\(&(flags|ids), vertices_count\);$",
     r"\1#ifdef __USE_HIP__" "\n" r"\1MemoryAPI::allocate_device_array(&\2, vertices_count);" "\n" r"\1#else" "\n" r"\1MemoryAPI::allocate_array(&\2, vertices_count);" "\n" r"\1#endif", 2),
    ("vgl_datastructures/frontier/containers/vect_csr/frontier_vect_csr.hpp", "sub", r"^(\s*)MemoryAPI::allocate_array\(&(flags|ids), max_size\);$",
     r"\1#ifdef __USE_HIP__" "\n" r"\1MemoryAPI::allocate_device_array(&\2, max_size);" "\n" r"\1#else" "\n" r"\1MemoryAPI::allocate_array(&\2, max_size);" "\n" r"\1#endif", 2),
    # a graph container that frees its arrays tells the backend (device copies of the adjacency are keyed by the container's address)
    ("vgl_datastructures/graphs/undirected_containers/csr/csr_graph.hpp", "sub", r"^(void CSRGraph::free\(\)\n\{\n)",
     r"\1    #ifdef __USE_HIP__\n    hip_container_changed(this);\n    #endif\n", 1),
    ("vgl_datastructures/graphs/undirected_containers/vect_csr/vect_csr_graph.hpp", "sub", r"^(void VectorCSRGraph::free\(\)\n\{\n)",
     r"\1    #ifdef __USE_HIP__\n    hip_container_changed(this);\n    #endif\n", 1),
    # ---- host-side container code that exists per architecture: take the plain C++ variant of the CUDA flavour ----
    ("vgl_datastructures/graphs/undirected_containers/edges_list/preprocess_into_segmented.hpp", "sub",
     r"^#ifdef __USE_GPU__$(?=\nvoid EdgesListGraph::preprocess_into_segmented)", GPU_OR_HIP),
    ("vgl_runtime/helpers/sorter/sorter.h", "sub", r"^(\s*)#ifdef __USE_MULTICORE__$", r"\1#if defined(__USE_MULTICORE__) || defined(__USE_HIP__)"),
    ("vgl_datastructures/graphs/undirected_containers/csr/reorder.hpp", "sub", r"defined\(__USE_NEC_SX_AURORA__\) \|\| defined\(__USE_MULTICORE__\)$",
     "defined(__USE_NEC_SX_AURORA__) || defined(__USE_MULTICORE__) || defined(__USE_HIP__)", 2),
    ("vgl_datastructures/graphs/undirected_containers/vect_csr/reorder.hpp", "sub", r"defined\(__USE_NEC_SX_AURORA__\) \|\| defined\(__USE_MULTICORE__\)$",
     "defined(__USE_NEC_SX_AURORA__) || defined(__USE_MULTICORE__) || defined(__USE_HIP__)", 2),
    # ---- a stamp on the frontier containers: generate_new_frontier of the HIP backend leaves the advance plan of a sparse frontier behind and marks
    #      the container; every mutator of the container voids the mark ("plan stamp") ----
    ("vgl_datastructures/frontier/containers/base_frontier.h", "after", r"^\s*FrontierSparsityType sparsity_type;$",
     "    #ifdef __USE_HIP__\n    unsigned long long hip_plan_token = 0;   // non-zero: ids / flags are as GraphAbstractionsHIP::generate_new_frontier left them\n    #endif\n"),
    ("vgl_datastructures/frontier/containers/csr/modification.hpp", "sub", r"^(void FrontierCSR::(?:set_all_active|add_vertex|clear)\([^)]*\)\n\{\n)",
     r"\1    #ifdef __USE_HIP__\n    hip_plan_token = 0;\n    #endif\n", 3),
    ("vgl_datastructures/frontier/containers/vect_csr/modification.hpp", "sub", r"^(void FrontierVectorCSR::(?:set_all_active|add_vertex|add_group_of_vertices)\([^)]*\)\n\{\n)",
     r"\1    #ifdef __USE_HIP__\n    hip_plan_token = 0;\n    #endif\n", 3),
    ("vgl_datastructures/frontier/containers/vect_csr/frontier_vect_csr.h", "sub", r"^(\s*void clear\(\) \{)(sparsity_type = SPARSE_FRONTIER;)",
     r"\1\n    #ifdef __USE_HIP__\n    hip_plan_token = 0;\n    #endif\n    \2", 1),
    # ---- the backend writes the frontier containers, like the other backends ----
    ("vgl_datastructures/frontier/containers/csr/frontier_csr.h", "after", r"friend class GraphAbstractionsGPU;", "    friend class GraphAbstractionsHIP;\n"),
    ("vgl_datastructures/frontier/containers/vect_csr/frontier_vect_csr.h", "after", r"friend class GraphAbstractionsGPU;", "    friend class GraphAbstractionsHIP;\n"),
    ("vgl_datastructures/frontier/containers/csr_vg/frontier_csr_vg.h", "after", r"friend class GraphAbstractionsGPU;", "    friend class GraphAbstractionsHIP;\n"),
    ("vgl_datastructures/frontier/containers/edges_list/frontier_edges_list.h", "after", r"friend class GraphAbstractionsGPU;", "    friend class GraphAbstractionsHIP;\n"),
]


def apply_tree_rule(root, rule):
    """("*", "sub_tree", regex, replacement, expected total, excluded files): every header and source below the tree except the CUDA backend"""
    _, _, anchor, text, expected, excluded = rule
    rx = re.compile(anchor, re.M)
    total = 0
    for d, _dirs, files in os.walk(root):
        rel = os.path.relpath(d, root)
        if rel.startswith(".git") or rel.startswith(os.path.join("vgl_compute_api", "gpu")):
            continue
        for name in files:
            if not name.endswith((".h", ".hpp", ".cpp")) or os.path.normpath(os.path.join(rel, name)) in excluded:
                continue
            full = os.path.join(d, name)
            with open(full, errors="ignore") as f:
                src = f.read()
            n = len(rx.findall(src))
            if n:
                with open(full, "w") as f:
                    f.write(rx.sub(text, src))
                total += n
    if total != expected:
        raise SystemExit(f"apply_hip_binding: tree rule /{anchor}/ matched {total} places, expected {expected}")


def apply_rule(root, rule):
    if rule[1] == "sub_tree":
        return apply_tree_rule(root, rule)
    path, kind, anchor, text = rule[:4]
    expected = rule[4] if len(rule) > 4 else 1
    which = rule[5] if len(rule) > 5 else 0
    full = os.path.join(root, path)
    with open(full) as f:
        src = f.read()
    rx = re.compile(anchor, re.M)
    found = len(rx.findall(src))
    if found != expected:
        raise SystemExit(f"apply_hip_binding: {path}: anchor /{anchor}/ found {found} times, expected {expected}")
    if kind == "sub":
        src = rx.sub(text, src)
    else:
        m = list(rx.finditer(src))[which]
        line_start = src.rfind("\n", 0, m.start()) + 1
        line_end = src.find("\n", m.end())
        line_end = len(src) if line_end < 0 else line_end + 1
        pos = line_start if kind == "before" else line_end
        src = src[:pos] + text + src[pos:]
    with open(full, "w") as f:
        f.write(src)


def main():
    if len(sys.argv) != 2 or not os.path.isfile(os.path.join(sys.argv[1], "graph_library.h")):
        raise SystemExit("usage: apply_hip_binding.py <directory holding a copy of the VGL source tree>")
    root = os.path.abspath(sys.argv[1])
    if os.path.exists(os.path.join(root, "vgl_compute_api", "hip")):
        raise SystemExit("apply_hip_binding: this tree already has vgl_compute_api/hip")
    for rule in RULES:
        apply_rule(root, rule)
    shutil.copytree(os.path.join(HERE, "vgl_compute_api", "hip"), os.path.join(root, "vgl_compute_api", "hip"))
    print(f"apply_hip_binding: {len(RULES)} edits applied, vgl_compute_api/hip added")


if __name__ == "__main__":
    main()
