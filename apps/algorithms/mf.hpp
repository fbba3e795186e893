// Maximum flow, Ford-Fulkerson with breadth-first augmenting paths (algorithms/mf/mf.hpp:5-128, seq_mf.hpp:51-97) on the operator
// API.  The reference's network semantics are kept: every stored edge carries its own residual value in an EdgesArray (initially
// MAX_WEIGHT), an augmentation subtracts from ALL parallel edges u->v and adds to ALL stored edges v->u (no reverse edge stored: nothing
// to add to), the value of u->v is read from the first match.  On a symmetric graph this is the textbook residual network and the
// result is the maximum flow; on a directed graph the result depends on the augmenting paths chosen.
// Differences from the reference, both to make a run reproducible: among the frontier vertices that can reach a vertex the parent is
// the smallest id (atomicMin) instead of the last writer, and the walk along the path (minimum, update) runs in one workgroup on the
// device instead of on the host through per-edge accessors.
#pragma once
#include "bfs.hpp"
#define MAX_WEIGHT 100
#define MF_NO_PARENT 0x7FFFFFFF

// one workgroup walks sink -> source along parents[]: result[0] = bottleneck, result[1] = path length (0 and -1 on a broken chain)
template <typename _T>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_mf_augment(vgl_csr_view out, _T *flows, const int *parents, int source, int sink, int vertices_count,
                                                              _T max_value, long long *result)
{
    __shared__ int first_match;
    _T path_flow = max_value;
    int length = 0;
    bool broken = false;
    for (int v = sink; v != source; length++) {               // parents[] is the same for every thread: uniform control flow
        const int u = parents[v];
        if (u < 0 || u >= vertices_count || length >= vertices_count) { broken = true; break; }
        const long long lo = out.rowptr[u]; const int deg = (int)(out.rowptr[u + 1] - lo);
        if (threadIdx.x == 0) first_match = 0x7FFFFFFF;
        __syncthreads();
        for (int p = threadIdx.x; p < deg; p += VGL_BLOCK) if (out.adj[lo + p] == v) atomicMin(&first_match, p);
        __syncthreads();
        const _T w = first_match == 0x7FFFFFFF ? (_T)0 : flows[lo + first_match];
        path_flow = w < path_flow ? w : path_flow;
        __syncthreads();
        v = u;
    }
    if (!broken && path_flow > 0)
        for (int v = sink; v != source;) {
            const int u = parents[v];
            const long long ulo = out.rowptr[u], uhi = out.rowptr[u + 1], vlo = out.rowptr[v], vhi = out.rowptr[v + 1];
            for (long long p = ulo + threadIdx.x; p < uhi; p += VGL_BLOCK) if (out.adj[p] == v) flows[p] -= path_flow;
            __syncthreads();                                  // u == v never happens on a BFS path, but the two loops may share a row with the next hop
            for (long long p = vlo + threadIdx.x; p < vhi; p += VGL_BLOCK) if (out.adj[p] == u) flows[p] += path_flow;
            __syncthreads();
            v = u;
        }
    if (threadIdx.x == 0) { result[0] = broken ? 0 : (long long)path_flow; result[1] = broken ? -1 : length; }
}

struct MaxFlow {
    template <typename _T>
    static bool mf_bfs(VGL_Graph &graph, EdgesArray<_T> &weights, int source, int sink, VerticesArray<int> &parents, VerticesArray<int> &levels,
                       VGL_GRAPH_ABSTRACTIONS &api, VGL_FRONTIER &front)
    {
        front.set_all_active();
        auto init = [parents, levels, source] __VGL_COMPUTE_ARGS__ {
            parents[src_id] = MF_NO_PARENT;
            levels[src_id] = src_id == source ? FIRST_LEVEL_VERTEX : UNVISITED_VERTEX;
        };
        api.compute(graph, front, init);
        front.clear();
        front.add_vertex(source);
        int current_level = FIRST_LEVEL_VERTEX;
        while (front.size() > 0) {
            auto edge_op = [levels, parents, weights, current_level] __VGL_ADVANCE_ARGS__ {
                const int dst_level = levels[dst_id];
                if ((dst_level == UNVISITED_VERTEX || dst_level == current_level + 1) && weights[global_edge_pos] > 0) {
                    levels[dst_id] = current_level + 1;
                    atomicMin(&parents[dst_id], src_id);
                }
            };
            api.scatter(graph, front, edge_op);
            auto on_next_level = [levels, current_level] __VGL_GNF_ARGS__ {
                return levels[src_id] == current_level + 1 ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG;
            };
            api.generate_new_frontier(graph, front, on_next_level);
            current_level++;
        }
        int sink_level = UNVISITED_VERTEX;
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), &sink_level, levels.get_ptr() + sink, sizeof(int)));
        return sink_level != UNVISITED_VERTEX;
    }

    template <typename _T>
    static double vgl_ford_fulkerson(VGL_Graph &graph, EdgesArray<_T> &flows, int source, int sink, _T &max_flow)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER front(graph);
        VerticesArray<int> parents(graph), levels(graph);
        api.change_traversal_direction(SCATTER, front, parents, levels);
        long long *d_result = nullptr;
        MemoryAPI::allocate_device_array(&d_result, 2);
        Timer tm;
        tm.start();
        long long path_total = 0; int iterations_count = 0;
        while (source != sink && mf_bfs(graph, flows, source, sink, parents, levels, api, front)) {
            hipLaunchKernelGGL(vgl_k_mf_augment<_T>, dim3(1), dim3(VGL_BLOCK), 0, VGL_RUNTIME::stream(), graph.get_direction_view(SCATTER), flows.get_ptr(),
                               (const int *)parents.get_ptr(), source, sink, graph.get_vertices_count(), std::numeric_limits<_T>::max(), d_result);
            VGL_HIP_RT(hipGetLastError());
            long long r[2] = {0, 0};
            VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), r, d_result, sizeof(r)));
            if (r[1] < 0) throw "Error in MF::vgl_ford_fulkerson : broken parent chain";
            if (r[0] <= 0) break;                                  // cannot happen after a successful search; guards the loop
            max_flow += (_T)r[0];
            path_total += r[1]; iterations_count++;
        }
        tm.end();
        MemoryAPI::free_device_array(d_result);
        std::cout << "iterations done: " << iterations_count << std::endl;
        std::cout << "average path length: " << (iterations_count ? path_total / iterations_count : 0) << std::endl;
        performance_stats.print_algorithm_performance_stats("MF (Ford-Fulkerson)", tm.get_time(), graph.get_edges_count());
        return performance_stats.get_algorithm_performance(tm.get_time(), graph.get_edges_count());
    }
};
#define MF MaxFlow
