// advance_contract: self-check of the six-functor advance contract of the operator API (common/advance.hpp:6-115,
// multicore/advance_worker.hpp:62-149,204-319) on the HIP backend, for both storage formats and the three frontier kinds.
//   CSR_GRAPH        : the collective functor set is never called (the reference's CSR worker does not call it either)
//   VECTOR_CSR_GRAPH : rows shorter than VECTOR_CORE_THRESHOLD_VALUE get the collective set, the others the primary set
// Every operator of a set counts into its own counter; the expected counts come from the row lengths on the host.
#define VECTOR_CORE_THRESHOLD_VALUE 24
#include "common.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(parser.format);
        prepare_graph(graph, parser);
        const int V = graph.get_vertices_count();
        int errors = 0;
        long long *cnt;                                         // [0] edge, [1] pre, [2] post, [3] collective edge, [4] collective pre, [5] collective post, [6] checksum
        MemoryAPI::allocate_array(&cnt, 8);
        for (int pass = 0; pass < 6; pass++) {
            const TraversalDirection dir = pass < 3 ? SCATTER : GATHER;
            VGL_GRAPH_ABSTRACTIONS api(graph, dir);
            VGL_FRONTIER frontier(graph, dir);
            VerticesArray<int> marks(graph, dir);
            api.attach_data(marks);
            api.change_traversal_direction(dir, marks, frontier);
            HostCSR h(graph, dir);
            // frontier kinds: all-active, sparse (every 7th vertex), dense for vcsr / sparse for csr (all but every 11th vertex)
            const int kind = pass % 3;
            frontier.set_all_active();
            auto pick = [marks, kind] __VGL_COMPUTE_ARGS__ { marks[src_id] = kind == 0 ? 1 : kind == 1 ? (src_id % 7 == 0) : (src_id % 11 != 0); };
            api.compute(graph, frontier, pick);
            if (kind != 0) { auto in_front = [marks] __VGL_GNF_ARGS__ { return marks[src_id]; }; api.generate_new_frontier(graph, frontier, in_front); }
            const FrontierSparsityType want = kind == 0 ? ALL_ACTIVE_FRONTIER : (kind == 2 && graph.get_format() == VECTOR_CSR_GRAPH) ? DENSE_FRONTIER : SPARSE_FRONTIER;
            if (frontier.get_sparsity_type() != want) { std::cout << "pass " << pass << ": unexpected frontier type " << (int)frontier.get_sparsity_type() << std::endl; errors++; }
            for (int i = 0; i < 8; i++) cnt[i] = 0;
            const long long shift = dir == GATHER ? graph.get_edges_count() : 0;
            const vgl_csr_view view = graph.get_direction_view(dir);
            auto edge = [cnt, view, shift] __VGL_ADVANCE_ARGS__ {
                atomicAdd((unsigned long long *)&cnt[0], 1ULL);
                // local_edge_pos / global_edge_pos contract: the edge is entry local_edge_pos of the row, global = shift + CSR position
                if (view.adj[global_edge_pos - shift] != dst_id || global_edge_pos - shift != view.rowptr[src_id] + local_edge_pos) atomicAdd((unsigned long long *)&cnt[6], 1ULL);
            };
            auto pre = [cnt] __VGL_ADVANCE_PREPROCESS_ARGS__ { atomicAdd((unsigned long long *)&cnt[1], 1ULL); };
            auto post = [cnt] __VGL_ADVANCE_POSTPROCESS_ARGS__ { atomicAdd((unsigned long long *)&cnt[2], (unsigned long long)connections_count); };
            auto c_edge = [cnt, view, shift] __VGL_ADVANCE_ARGS__ {
                atomicAdd((unsigned long long *)&cnt[3], 1ULL);
                if (view.adj[global_edge_pos - shift] != dst_id || global_edge_pos - shift != view.rowptr[src_id] + local_edge_pos) atomicAdd((unsigned long long *)&cnt[6], 1ULL);
            };
            auto c_pre = [cnt] __VGL_ADVANCE_PREPROCESS_ARGS__ { atomicAdd((unsigned long long *)&cnt[4], 1ULL); };
            auto c_post = [cnt] __VGL_ADVANCE_POSTPROCESS_ARGS__ { atomicAdd((unsigned long long *)&cnt[5], (unsigned long long)connections_count); };
            if (dir == SCATTER) api.scatter(graph, frontier, edge, pre, post, c_edge, c_pre, c_post);
            else api.gather(graph, frontier, edge, pre, post, c_edge, c_pre, c_post);
            long long want_cnt[6] = {0, 0, 0, 0, 0, 0};
            const bool split = graph.get_format() == VECTOR_CSR_GRAPH;
            for (int v = 0; v < V; v++) {
                const bool active = kind == 0 ? true : kind == 1 ? (v % 7 == 0) : (v % 11 != 0);
                if (!active) continue;
                const long long deg = h.rowptr[(size_t)v + 1] - h.rowptr[(size_t)v];
                const int set = (split && deg < VECTOR_CORE_THRESHOLD_VALUE) ? 3 : 0;
                want_cnt[set] += deg; want_cnt[set + 1] += 1; want_cnt[set + 2] += deg;
            }
            for (int i = 0; i < 6; i++)
                if (cnt[i] != want_cnt[i]) { std::cout << "pass " << pass << " counter " << i << ": " << cnt[i] << " vs " << want_cnt[i] << std::endl; errors++; }
            if (cnt[6] != 0) { std::cout << "pass " << pass << ": " << cnt[6] << " edges with wrong positions" << std::endl; errors++; }
            if (!split && (cnt[3] | cnt[4] | cnt[5])) errors++;
            // wrong direction must throw (common/advance.hpp:19-26)
            bool thrown = false;
            try { if (dir == SCATTER) api.gather(graph, frontier, edge); else api.scatter(graph, frontier, edge); } catch (const char *) { thrown = true; }
            if (!thrown) { std::cout << "pass " << pass << ": direction mismatch not refused" << std::endl; errors++; }
        }
        MemoryAPI::free_array(cnt);
        std::cout << "error count: " << errors << std::endl;
        VGL_RUNTIME::finalize_library();
        return errors ? 1 : 0;
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
