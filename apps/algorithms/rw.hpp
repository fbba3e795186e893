// Random walks (algorithms/rw/random_walk.hpp:5-85): every walk vertex starts a walk at itself; per step one compute() over the
// sparse frontier of walk vertices moves each walk to a random out-neighbour of its current vertex, or to DEAD_END where there is none.
// The reference draws rand() values into an array per step; here the draw is a counter-based hash of (seed, step, ORIGINAL id of the
// walk vertex), so a run is reproducible whatever the storage format -- which is what lets a test check it against a CPU restatement
// (the reference's own app gives up on checking: apps/rw/rw.cpp "since walks are random it is not possible to check").
// The lambda reads the adjacency through a vgl_csr_view captured by value (the reference captures the host graph object by
// reference, which is why it restricts this algorithm to its CPU backends).
#pragma once
#define DEAD_END -1

struct RandomWalk {
    __host__ __device__ static inline unsigned long long draw(unsigned long long seed, unsigned long long step, unsigned long long walk)
    {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (step + 1) + 0xD1B54A32D192ED03ULL * (walk + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    // walk vertices: ORIGINAL ids v with draw(seed, all-ones, v) % 100 < percent (apps/rw/rw.cpp:27-33)
    __host__ __device__ static inline bool is_walk_vertex(unsigned long long seed, int original_id, int percent)
    { return (int)(draw(seed, ~0ULL, (unsigned long long)original_id) % 100ULL) < percent; }

    template <typename _T>
    static double vgl_random_walk(VGL_Graph &graph, int walk_vertices_percent, int walk_length, unsigned long long seed, VerticesArray<_T> &walk_results)
    {
        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER front(graph, SCATTER);
        api.change_traversal_direction(SCATTER, front, walk_results);
        const vgl_csr_view out = graph.get_direction_view(SCATTER);
        const int *to_original = graph.get_backward_conversion();        // nullptr: stored ids are the original ids
        Timer tm;
        tm.start();
        walk_results.set_all_constant(DEAD_END);
        auto walk_vertex = [to_original, seed, walk_vertices_percent] __VGL_GNF_ARGS__ {
            return is_walk_vertex(seed, to_original ? to_original[src_id] : src_id, walk_vertices_percent) ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG;
        };
        api.generate_new_frontier(graph, front, walk_vertex);
        auto init_walks = [walk_results] __VGL_COMPUTE_ARGS__ { walk_results[src_id] = src_id; };
        api.compute(graph, front, init_walks);
        for (int iteration = 0; iteration < walk_length; iteration++) {
            auto visit_next = [walk_results, out, to_original, seed, iteration] __VGL_COMPUTE_ARGS__ {
                const int current_id = walk_results[src_id];
                if (current_id == DEAD_END) return;
                const long long first = out.rowptr[current_id];
                const long long current_connections_count = out.rowptr[current_id + 1] - first;
                if (current_connections_count > 0) {
                    const unsigned long long r = draw(seed, (unsigned long long)iteration, (unsigned long long)(to_original ? to_original[src_id] : src_id));
                    walk_results[src_id] = out.adj[first + (long long)(r % (unsigned long long)current_connections_count)];
                } else walk_results[src_id] = DEAD_END;
            };
            api.compute(graph, front, visit_next);
        }
        tm.end();
        std::cout << "walk vertices num: " << front.size() << std::endl << "walk length: " << walk_length << std::endl;
        // one adjacency read per walk and step
        const double steps = (double)front.size() * walk_length;
        return tm.get_time() > 0 ? steps / (tm.get_time() * 1e6) : 0.0;
    }
};
#define RW RandomWalk
