// sssp_convergence_check -- a program against the VGL tree WITH the HIP binding applied (oracle/Makefile, target `binding`; not part of the reference).
//
// The reference's GPU Bellman-Ford (algorithms/sssp/gpu_shortest_paths.hpp:29-57) ends when a whole pass over the edges leaves `changes[0]` at 0,
// a one-word flag its device lambdas store to and the host reads between passes.  This program runs that loop (the same operator, written out here)
// with the flag in three kinds of memory -- MemoryAPI::allocate_array (what the reference's code gets), pinned coherent host memory, device memory
// read back with a copy -- and after the loop has ended runs EXTRA passes: a pass that still reports a change, or distances that differ from the
// sequential Dijkstra of the reference, mean the loop ended early (a store to the flag that the host did not see).
//
// usage: vgl_hip_sssp_convergence_check -load <graph file> | -s <scale> -e <edge factor> -type rmat, -format csr|vcsr, -it <sources>

#define INT_ELEMENTS_PER_EDGE 5.0
#define VECTOR_ENGINE_THRESHOLD_VALUE VECTOR_LENGTH*MAX_SX_AURORA_THREADS*128
#define VECTOR_CORE_THRESHOLD_VALUE 5*VECTOR_LENGTH

#include "graph_library.h"

enum FlagMemory { FLAG_MEMORY_API = 0, FLAG_PINNED_COHERENT = 1, FLAG_DEVICE = 2 };
static const char *flag_name[] = {"MemoryAPI::allocate_array", "hipHostMalloc(coherent)", "device memory + copy"};
// how the operator touches `distances`: as the reference writes it (plain loads, plain conditional store), with the store as a device-scope
// atomic minimum on the bit pattern (non-negative floats order like integers), or with every access at agent scope (sc1 loads and stores)
enum OperatorForm { OP_ATOMIC_MIN = 0, OP_PLAIN = 1, OP_AGENT_SCOPE = 2, OP_AGENT_STORE = 3 };
static const char *op_name[] = {"store = atomicMin", "plain loads and stores (the reference's operator)", "agent-scope loads and stores", "plain loads, agent-scope stores"};

static int read_flag(int *flag, int kind)
{
    if (kind != FLAG_DEVICE) return flag[0];
    int v = 0;
    if (hipMemcpy(&v, flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) throw "hipMemcpy failed";
    return v;
}
static void clear_flag(int *flag, int kind)
{
    if (kind != FLAG_DEVICE) { flag[0] = 0; return; }
    if (hipMemset(flag, 0, sizeof(int)) != hipSuccess) throw "hipMemset failed";
}

int main(int argc, char **argv)
{
    int failures = 0;
    try
    {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        parser.parse_args(argc, argv);
        VGL_Graph graph(VGL_RUNTIME::select_graph_format(parser), VGL_RUNTIME::select_graph_optimizations(parser));
        VGL_RUNTIME::prepare_graph(graph, parser);

        VerticesArray<float> distances(graph, SCATTER), check_distances(graph, SCATTER);
        EdgesArray<float> weights(graph);
        weights.set_all_random(MAX_WEIGHT);

        for (int round = 0; round < parser.get_number_of_rounds(); round++)
        {
            const int source_vertex = graph.select_random_nz_vertex(SCATTER);
            // the sequential Dijkstra of the reference checks the atomic form for the first two sources; every other run is compared with the atomic form's result
            if (round < 2) ShortestPaths::seq_dijkstra(graph, weights, check_distances, source_vertex);
            for (int form = 0; form < 4; form++)
            {
                const int kind = FLAG_DEVICE;
                VGL_GRAPH_ABSTRACTIONS graph_API(graph, SCATTER);
                VGL_Frontier frontier(graph, SCATTER);
                graph_API.change_traversal_direction(SCATTER, distances, frontier);
                weights.move_to_device();
                distances.move_to_device();
                int *changes = NULL;
                if (kind == FLAG_MEMORY_API) MemoryAPI::allocate_array(&changes, 1);
                else if (kind == FLAG_PINNED_COHERENT) { if (hipHostMalloc((void **)&changes, sizeof(int), hipHostMallocCoherent) != hipSuccess) throw "hipHostMalloc failed"; }
                else if (hipMalloc((void **)&changes, sizeof(int)) != hipSuccess) throw "hipMalloc failed";

                const float inf_val = std::numeric_limits<float>::max() - MAX_WEIGHT;
                auto init_op = [distances, source_vertex, inf_val] __VGL_COMPUTE_ARGS__ {
                    distances[src_id] = src_id == source_vertex ? 0 : inf_val;
                };
                frontier.set_all_active();
                graph_API.compute(graph, frontier, init_op);
                auto edge_op = [weights, distances, changes] __VGL_SCATTER_ARGS__ {
                    float weight = weights[global_edge_pos];
                    float src_weight = distances[src_id];
                    float dst_weight = distances[dst_id];
                    if (dst_weight > src_weight + weight) { distances[dst_id] = src_weight + weight; changes[0] = 1; }
                };
                auto edge_op_atomic = [weights, distances, changes] __VGL_SCATTER_ARGS__ {
                    float weight = weights[global_edge_pos];
                    float src_weight = distances[src_id];
                    float dst_weight = distances[dst_id];
                    if (dst_weight > src_weight + weight)
                        if (atomicMin((int *)&distances[dst_id], __float_as_int(src_weight + weight)) > __float_as_int(src_weight + weight)) changes[0] = 1;
                };
                auto edge_op_agent = [weights, distances, changes] __VGL_SCATTER_ARGS__ {
                    float weight = weights[global_edge_pos];
                    float src_weight = __hip_atomic_load(&distances[src_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    float dst_weight = __hip_atomic_load(&distances[dst_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (dst_weight > src_weight + weight) { __hip_atomic_store(&distances[dst_id], src_weight + weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); changes[0] = 1; }
                };
                auto edge_op_agent_store = [weights, distances, changes] __VGL_SCATTER_ARGS__ {
                    float weight = weights[global_edge_pos];
                    float src_weight = distances[src_id];
                    float dst_weight = distances[dst_id];
                    if (dst_weight > src_weight + weight) { __hip_atomic_store(&distances[dst_id], src_weight + weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); changes[0] = 1; }
                };
                int passes = 0, extra_changes = 0;
                auto pass = [&]() {
                    if (form == OP_PLAIN) graph_API.scatter(graph, frontier, edge_op);
                    else if (form == OP_ATOMIC_MIN) graph_API.scatter(graph, frontier, edge_op_atomic);
                    else if (form == OP_AGENT_SCOPE) graph_API.scatter(graph, frontier, edge_op_agent);
                    else graph_API.scatter(graph, frontier, edge_op_agent_store);
                };
                do { clear_flag(changes, kind); pass(); passes++; } while (read_flag(changes, kind) > 0);
                for (int extra = 0; extra < 3; extra++) { clear_flag(changes, kind); pass(); extra_changes += read_flag(changes, kind) > 0; }
                long long wrong = 0;
                const int vertices_count = graph.get_vertices_count();
                VerticesArray<float> got(graph, SCATTER);
                for (int v = 0; v < vertices_count; v++) got[v] = distances[v];
                got.reorder(ORIGINAL);
                check_distances.reorder(ORIGINAL);
                if (form == OP_ATOMIC_MIN && round >= 2) for (int v = 0; v < vertices_count; v++) check_distances[v] = got[v];
                for (int v = 0; v < vertices_count; v++) wrong += got[v] != check_distances[v];
                check_distances.reorder(SCATTER);
                cout << "source " << source_vertex << ", " << op_name[form] << ": " << passes << " passes, " << extra_changes
                     << " of 3 extra passes still changed something, " << wrong << " distances differ from "
                     << (round < 2 ? "seq_dijkstra" : "the atomic form") << endl;
                failures += (extra_changes != 0 || wrong != 0);
                if (kind == FLAG_MEMORY_API) MemoryAPI::free_array(changes);
                else if (kind == FLAG_PINNED_COHERENT) (void)hipHostFree(changes);
                else (void)hipFree(changes);
            }
        }
        cout << (failures ? "SSSP CONVERGENCE CHECK FAILED" : "SSSP CONVERGENCE CHECK PASSED") << endl;
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 2; }
    catch (const char *error) { cout << error << endl; return 2; }
    return failures ? 1 : 0;
}
