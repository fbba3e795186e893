// plan_stamp_check -- a program against the VGL tree WITH the HIP binding applied (oracle/Makefile, target `binding`; not part of the reference).
//
// GraphAbstractionsHIP::generate_new_frontier leaves the advance plan of a sparse frontier behind and stamps the container
// (BaseFrontier::hip_plan_token); the next scatter starts from that plan.  Host code may rewrite the frontier in between -- clear / add_vertex /
// add_group_of_vertices / set_all_active -- and the container's mutators void the stamp.  Each case below runs one scatter on a frontier that was
// GENERATED and then (cases 2, 3) rewritten by the host, and the same scatter on a frontier object the backend has never generated (no stamp: the
// plan is rebuilt from the ids); the vertices the two reach must be the same.  The rewritten frontier has the SIZE of the generated one and other
// degrees, so a plan that survived the rewrite sends the scatter over the wrong edge ranges.
//
// usage: vgl_hip_plan_stamp_check -s <scale> -e <edge factor> -type rmat -format csr|vcsr          prints PLAN STAMP CHECK PASSED / FAILED

#define INT_ELEMENTS_PER_EDGE 4.0
#define NEC_VECTOR_ENGINE_THRESHOLD_VALUE  VECTOR_LENGTH * MAX_SX_AURORA_THREADS * 128
#define VECTOR_CORE_THRESHOLD_VALUE 2*VECTOR_LENGTH
#define COLLECTIVE_FRONTIER_TYPE_CHANGE_THRESHOLD 0.35

#include "graph_library.h"

static void put(VGL_Graph &graph, VGL_FRONTIER &frontier, int a, int b)
{
    frontier.clear();
    if (graph.get_container_type() == VECTOR_CSR_GRAPH) { int pair[2] = {a, b}; frontier.add_group_of_vertices(pair, 2); }   // (add_vertex takes one vertex there)
    else { frontier.add_vertex(a); frontier.add_vertex(b); }
}

static void reach(VGL_Graph &graph, VGL_GRAPH_ABSTRACTIONS &api, VGL_FRONTIER &frontier, VerticesArray<int> &out)
{
    const int vertices_count = graph.get_vertices_count();
    for (int v = 0; v < vertices_count; v++) out[v] = 0;
    auto touch = [out] __VGL_SCATTER_ARGS__ { out[dst_id] = 1; };
    api.scatter(graph, frontier, touch);
}

static long long differences(VGL_Graph &graph, VerticesArray<int> &a, VerticesArray<int> &b, long long *reached)
{
    long long diff = 0; *reached = 0;
    for (int v = 0; v < graph.get_vertices_count(); v++) { diff += a[v] != b[v]; *reached += a[v]; }
    return diff;
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
        const int vertices_count = graph.get_vertices_count();

        VGL_GRAPH_ABSTRACTIONS api(graph);
        VGL_FRONTIER frontier(graph), fresh(graph);
        VerticesArray<int> mark(graph, SCATTER), got(graph, SCATTER), want(graph, SCATTER), hubs_reach(graph, SCATTER);
        api.change_traversal_direction(SCATTER, mark, got, want, hubs_reach, frontier, fresh);

        // two vertices of the largest degrees and two of the smallest non-zero ones (ids of the SCATTER direction)
        int hub[2] = {-1, -1}, leaf[2] = {-1, -1};
        for (int v = 0; v < vertices_count; v++) {
            const int d = graph.get_connections_count(v, SCATTER);
            if (d == 0) continue;
            if (hub[0] < 0 || d > graph.get_connections_count(hub[0], SCATTER)) { hub[1] = hub[0]; hub[0] = v; }
            else if (hub[1] < 0 || d > graph.get_connections_count(hub[1], SCATTER)) hub[1] = v;
            if (leaf[0] < 0 || d < graph.get_connections_count(leaf[0], SCATTER)) { leaf[1] = leaf[0]; leaf[0] = v; }
            else if (leaf[1] < 0 || d < graph.get_connections_count(leaf[1], SCATTER)) leaf[1] = v;
        }
        if (hub[1] < 0 || leaf[1] < 0 || hub[0] == leaf[0] || hub[0] == leaf[1] || hub[1] == leaf[0] || hub[1] == leaf[1]) throw "plan_stamp_check: the graph is too small";
        if (hub[0] > hub[1]) std::swap(hub[0], hub[1]);
        if (leaf[0] > leaf[1]) std::swap(leaf[0], leaf[1]);
        cout << "hubs " << hub[0] << " (" << graph.get_connections_count(hub[0], SCATTER) << " edges) " << hub[1] << " (" << graph.get_connections_count(hub[1], SCATTER)
             << "), leaves " << leaf[0] << " (" << graph.get_connections_count(leaf[0], SCATTER) << ") " << leaf[1] << " (" << graph.get_connections_count(leaf[1], SCATTER) << ")" << endl;

        for (int v = 0; v < vertices_count; v++) mark[v] = (v == hub[0] || v == hub[1]) ? 1 : 0;
        auto marked = [mark] __VGL_GNF_ARGS__ { return mark[src_id] == 1 ? IN_FRONTIER_FLAG : NOT_IN_FRONTIER_FLAG; };
        long long reached = 0, hub_reached = 0, diff = 0;

        // case 1: the generated frontier as it stands (the scatter starts from the plan the generation left)
        api.generate_new_frontier(graph, frontier, marked);
        if (frontier.size() != 2) throw "plan_stamp_check: the generated frontier does not hold the two marked vertices";
        reach(graph, api, frontier, got);
        put(graph, fresh, hub[0], hub[1]);
        reach(graph, api, fresh, want);
        diff = differences(graph, got, want, &hub_reached);
        cout << "case 1 (generated, untouched): " << hub_reached << " vertices reached, " << diff << " differences" << endl;
        failures += diff != 0 || hub_reached == 0;
        for (int v = 0; v < vertices_count; v++) hubs_reach[v] = got[v];

        // case 2: generated, then rewritten by the host to two other vertices -- same size, other degrees
        api.generate_new_frontier(graph, frontier, marked);
        put(graph, frontier, leaf[0], leaf[1]);
        reach(graph, api, frontier, got);
        put(graph, fresh, leaf[0], leaf[1]);
        reach(graph, api, fresh, want);
        diff = differences(graph, got, want, &reached);
        cout << "case 2 (generated, then clear + add): " << reached << " vertices reached, " << diff << " differences" << endl;
        failures += diff != 0 || reached == 0;
        long long same_as_hubs = differences(graph, got, hubs_reach, &reached);
        failures += same_as_hubs == 0;                      // (the check must be able to tell the two frontiers apart)

        // case 3: generated, then set all-active by the host
        api.generate_new_frontier(graph, frontier, marked);
        frontier.set_all_active();
        reach(graph, api, frontier, got);
        fresh.set_all_active();
        reach(graph, api, fresh, want);
        diff = differences(graph, got, want, &reached);
        cout << "case 3 (generated, then set_all_active): " << reached << " vertices reached, " << diff << " differences" << endl;
        failures += diff != 0 || reached <= hub_reached;

        // case 4: generated twice in a row with different conditions (the second generation replaces plan and stamp)
        api.generate_new_frontier(graph, frontier, marked);
        for (int v = 0; v < vertices_count; v++) mark[v] = (v == leaf[0] || v == leaf[1]) ? 1 : 0;
        api.generate_new_frontier(graph, frontier, marked);
        reach(graph, api, frontier, got);
        put(graph, fresh, leaf[0], leaf[1]);
        reach(graph, api, fresh, want);
        diff = differences(graph, got, want, &reached);
        cout << "case 4 (generated again on other marks): " << reached << " vertices reached, " << diff << " differences" << endl;
        failures += diff != 0 || reached == 0;

        cout << (failures ? "PLAN STAMP CHECK FAILED" : "PLAN STAMP CHECK PASSED") << endl;
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 2; }
    catch (const char *error) { cout << error << endl; return 2; }
    return failures ? 1 : 0;
}
