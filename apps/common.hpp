// common.hpp -- shared pieces of the example applications: command line (cmd_parser.hpp:51-233 subset), graph
// preparation (VGL_RUNTIME::prepare_graph, vgl_runtime.hpp:27-60), host-side sequential checkers in the spirit of the
// reference's -check mode (seq_bfs.hpp, seq_shortest_paths.hpp, seq_pr.hpp, seq_bfs_based.hpp + verify_results.h), and
// the AVG_PERF / error count output lines the reference's harness greps (scripts/benchmarking_api.py:38-63).
#pragma once
#include "../vectorgraphlibrary_amd/hip/vgl_hip.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <queue>

struct Parser {
    int scale = 10, avg_degree = 5, rounds = 1, source = -1, sink = -1, walk_vertices_percent = 1;
    bool rmat = true, check = false, direction_optimising = false, fused = false, undirected = false, bfs_based = false, blocked = false, deterministic = false, declared = false;
    enum Traversal { PUSH_TRAVERSAL, PULL_TRAVERSAL } traversal = PUSH_TRAVERSAL;                   // cmd_parser.hpp (-push / -pull)
    enum FrontierKind { ALL_ACTIVE_KIND, PARTIAL_ACTIVE_KIND } frontier_kind = ALL_ACTIVE_KIND;     // (-all-active / -partial-active)
    GraphStorageFormat format = CSR_GRAPH;      // -format csr | vcsr (VECTOR_CSR_GRAPH: degree-renumbered, the reference's default)
    unsigned long long seed = 1;
    std::string dump, graph_file_name;
    enum ComputeMode { GENERATE_NEW_GRAPH, LOAD_GRAPH_FROM_FILE, IMPORT_EDGES_CONTAINER } compute_mode = GENERATE_NEW_GRAPH;   // cmd_parser.hpp:58-68
    void parse_args(int argc, char **argv)
    {
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto next = [&]() -> const char * { if (i + 1 >= argc) throw "missing value for command line option"; return argv[++i]; };
            if (a == "-s") scale = atoi(next());
            else if (a == "-e") avg_degree = atoi(next());
            else if (a == "-type") rmat = std::string(next()) == "rmat";
            else if (a == "-it") rounds = atoi(next());
            else if (a == "-check") check = true;
            else if (a == "-seed") seed = strtoull(next(), nullptr, 10);
            else if (a == "-dump") dump = next();
            else if (a == "-source") source = atoi(next());
            else if (a == "-sink") sink = atoi(next());
            else if (a == "-undirected") undirected = true;
            else if (a == "-import") { graph_file_name = next(); compute_mode = IMPORT_EDGES_CONTAINER; }     // .el_container
            else if (a == "-load" || a == "-file" || a == "-f") { graph_file_name = next(); compute_mode = LOAD_GRAPH_FROM_FILE; }   // .csr / .vcsr graph file
            else if (a == "-gen" || a == "-generate") compute_mode = GENERATE_NEW_GRAPH;
            else if (a == "-walk-vertices" || a == "-wv") walk_vertices_percent = atoi(next());     // cmd_parser.hpp:223-226
            else if (a == "-do") direction_optimising = true;
            else if (a == "-td") direction_optimising = false;
            else if (a == "-fused") fused = true;
            else if (a == "-declared") declared = true;                     // cc: the hook as a declared operator (VGL_MIN_LABEL_OVER_EDGES, API extension)
            else if (a == "-deterministic") deterministic = true;           // pr: sums in adjacency order, one lane per vertex (no float atomics)
            else if (a == "-blocked") blocked = true;                       // bfs -fused: prepare the graph for blocked top-down levels (once, not timed)
            else if (a == "-format") {
                const std::string f = next();
                format = (f == "vcsr" || f == "vect_csr") ? VECTOR_CSR_GRAPH : f == "el_container" ? EDGES_CONTAINER : CSR_GRAPH;
            }
            else if (a == "-bfs-based") bfs_based = true;            // tc: one BFS per source instead of Purdom's; cc: accepted
            else if (a == "-directed") undirected = false;
            // algorithm selectors of the reference's harness (apps/scripts/settings.py:15-25) that name what this backend does anyway
            else if (a == "-push") traversal = PUSH_TRAVERSAL;              // honoured by sssp (the other apps have one traversal, as in the reference)
            else if (a == "-pull") traversal = PULL_TRAVERSAL;
            else if (a == "-all-active") frontier_kind = ALL_ACTIVE_KIND;
            else if (a == "-partial-active") frontier_kind = PARTIAL_ACTIVE_KIND;
            else if (a == "-top-down" || a == "-cv" || a == "-purdoms") {}  // name what the app does anyway
            else throw "unknown command line option";
        }
    }
    int get_number_of_rounds() const { return rounds; }
    bool get_check_flag() const { return check; }
};

inline void prepare_graph(VGL_Graph &graph, const Parser &p, DirectionType dir = DIRECTED_GRAPH)
{
    GraphGenerationAPI::seed() = p.seed;
    EdgesContainer ec;
    if (p.compute_mode == Parser::LOAD_GRAPH_FROM_FILE) {              // vgl_runtime.hpp:52-60
        if (!graph.load_from_binary_file(p.graph_file_name)) throw "Error: graph file not found";
        return;
    }
    if (p.compute_mode == Parser::IMPORT_EDGES_CONTAINER) {            // vgl_runtime.hpp:61-75
        if (!ec.load_from_binary_file(p.graph_file_name)) throw "Error: edges container file not found";
        graph.import(ec);
        return;
    }
    const int v = 1 << p.scale;
    const long long e = (long long)v * p.avg_degree;
    if (p.rmat) GraphGenerationAPI::R_MAT(ec, v, e, 57, 19, 19, 5, dir);      // vgl_runtime.hpp:36
    else GraphGenerationAPI::random_uniform(ec, v, e, dir);
    graph.import(ec);
}

// -source / -sink are ORIGINAL vertex ids: refuse ids outside the graph before they index anything (host conversion tables, checkers)
inline int checked_vertex(const VGL_Graph &graph, int v, const char *what)
{
    if (v < 0 || v >= graph.get_vertices_count()) throw (std::string("Error: ") + what + " vertex id is outside [0, vertices count)");
    return v;
}

struct HostCSR {
    int V; std::vector<long long> rowptr; std::vector<int> adj;
    HostCSR(VGL_Graph &g, TraversalDirection d = SCATTER)
    {
        const vgl_csr_view v = g.get_direction_view(d);
        V = g.get_vertices_count();
        rowptr.resize((size_t)V + 1); adj.resize((size_t)v.edges);
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), rowptr.data(), v.rowptr, rowptr.size() * sizeof(long long)));
        VGL_HIP_CALL(vgl_hip_memcpy_d2h(VGL_RUNTIME::ctx(), adj.data(), v.adj, adj.size() * sizeof(int)));
    }
};

template <class T>
inline void dump_array(const std::string &path, const std::vector<T> &a)
{
    if (path.empty()) return;
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)a.data(), (std::streamsize)(a.size() * sizeof(T)));
}

// VGL_RUNTIME::report_performance + stop_measuring_stats (vgl_runtime.hpp): the line the harness greps, then -- when operator-API
// primitives ran -- the per-abstraction timers and the bandwidth under VGL's byte accounting (settings.h:140-155)
inline void report_performance(double mteps)
{
    std::cout << "AVG_PERF: " << mteps << " MTEPS" << std::endl;
    if (performance_stats.inner_wall_time > 0) performance_stats.print_timers_stats();
}

// verify_results (verify_results.h:33-92): exact for integers, |a-b| <= 100*FLT_EPSILON for floats
template <class T>
inline int verify_results(const std::vector<T> &a, const std::vector<T> &b)
{
    int errors = 0;
    for (size_t i = 0; i < a.size(); i++) {
        bool same;
        if (std::is_floating_point<T>::value) same = std::fabs((double)a[i] - (double)b[i]) <= 100.0 * FLT_EPSILON * std::max(1.0, std::fabs((double)b[i]));
        else same = a[i] == b[i];
        if (!same && errors++ < 10) std::cout << "error at " << i << ": " << a[i] << " vs " << b[i] << std::endl;
    }
    std::cout << "error count: " << errors << std::endl;
    return errors;
}

// ---- sequential checkers ----
inline std::vector<int> seq_bfs(const HostCSR &g, int source)
{
    std::vector<int> lv((size_t)g.V, -1);
    std::queue<int> q;
    lv[source] = 1; q.push(source);
    while (!q.empty()) {
        const int s = q.front(); q.pop();
        for (long long p = g.rowptr[s]; p < g.rowptr[s + 1]; p++)
            if (lv[g.adj[p]] == -1) { lv[g.adj[p]] = lv[s] + 1; q.push(g.adj[p]); }
    }
    return lv;
}
inline std::vector<float> seq_dijkstra(const HostCSR &g, const std::vector<float> &w, int source)
{
    const float inf = std::numeric_limits<float>::max() - MAX_WEIGHT;
    std::vector<float> d((size_t)g.V, inf);
    typedef std::pair<float, int> item;
    std::priority_queue<item, std::vector<item>, std::greater<item>> pq;
    d[source] = 0; pq.push({0.0f, source});
    while (!pq.empty()) {
        const int u = pq.top().second; pq.pop();
        for (long long p = g.rowptr[u]; p < g.rowptr[u + 1]; p++) {
            const int v = g.adj[p];
            if (d[v] > d[u] + w[p]) { d[v] = d[u] + w[p]; pq.push({d[v], v}); }
        }
    }
    return d;
}
// SCC::seq_tarjan (seq_scc.hpp): iterative Tarjan; labels are component counters (equal_components compares partitions)
inline std::vector<int> seq_tarjan(const HostCSR &g)
{
    const int V = g.V;
    std::vector<int> disc((size_t)V, -1), low((size_t)V, 0), comp((size_t)V, -1), stk, call_v;
    std::vector<long long> call_p;
    std::vector<char> onstk((size_t)V, 0);
    int timer = 0, ncomp = 0;
    for (int root = 0; root < V; root++) {
        if (disc[root] != -1) continue;
        call_v.push_back(root); call_p.push_back(g.rowptr[root]);
        disc[root] = low[root] = timer++; stk.push_back(root); onstk[root] = 1;
        while (!call_v.empty()) {
            const int u = call_v.back();
            if (call_p.back() < g.rowptr[u + 1]) {
                const int w = g.adj[call_p.back()++];
                if (disc[w] == -1) { disc[w] = low[w] = timer++; stk.push_back(w); onstk[w] = 1; call_v.push_back(w); call_p.push_back(g.rowptr[w]); }
                else if (onstk[w]) low[u] = std::min(low[u], disc[w]);
            } else {
                call_v.pop_back(); call_p.pop_back();
                if (!call_v.empty()) low[call_v.back()] = std::min(low[call_v.back()], low[u]);
                if (low[u] == disc[u]) {
                    for (;;) { const int x = stk.back(); stk.pop_back(); onstk[x] = 0; comp[x] = ncomp; if (x == u) break; }
                    ncomp++;
                }
            }
        }
    }
    return comp;
}
// HITS::seq_hits (hits.hpp:103-173): authorities from the incoming lists, hubs from the outgoing lists, 2-norm after each half step
inline void seq_hits(const HostCSR &out, const HostCSR &in, int steps, std::vector<double> &auth, std::vector<double> &hub)
{
    const int V = out.V;
    auth.assign((size_t)V, 1.0); hub.assign((size_t)V, 1.0);
    for (int step = 0; step < steps; step++) {
        double norm = 0;
        for (int v = 0; v < V; v++) {
            double acc = 0;
            for (long long p = in.rowptr[v]; p < in.rowptr[v + 1]; p++) acc += hub[in.adj[p]];
            auth[v] = acc; norm += acc * acc;
        }
        norm = std::sqrt(norm);
        for (int v = 0; v < V; v++) auth[v] /= norm;
        norm = 0;
        for (int v = 0; v < V; v++) {
            double acc = 0;
            for (long long p = out.rowptr[v]; p < out.rowptr[v + 1]; p++) acc += auth[out.adj[p]];
            hub[v] = acc; norm += acc * acc;
        }
        norm = std::sqrt(norm);
        for (int v = 0; v < V; v++) hub[v] /= norm;
    }
}
// in the spirit of SSWP::seq_dijkstra (seq_widest_paths.hpp:5-64): label-correcting with a max-priority queue
inline std::vector<float> seq_widest_paths(const HostCSR &g, const std::vector<float> &cap, int source)
{
    std::vector<float> wd((size_t)g.V, 0.0f);
    typedef std::pair<float, int> item;
    std::priority_queue<item> pq;
    wd[source] = std::numeric_limits<float>::max(); pq.push({wd[source], source});
    while (!pq.empty()) {
        const int u = pq.top().second; pq.pop();
        for (long long p = g.rowptr[u]; p < g.rowptr[u + 1]; p++) {
            const int v = g.adj[p];
            const float nw = std::min(wd[u], cap[p]);
            if (nw > wd[v]) { wd[v] = nw; pq.push({nw, v}); }
        }
    }
    return wd;
}
inline std::vector<float> seq_page_rank(const HostCSR &g, int iterations)
{
    const int V = g.V;
    const float d = 0.85f, k = (float)((1.0 - d) / ((float)V));
    std::vector<int> indeg((size_t)V, 0);
    for (int u = 0; u < V; u++) for (long long p = g.rowptr[u]; p < g.rowptr[u + 1]; p++) if (g.adj[p] != u) indeg[g.adj[p]]++;
    std::vector<float> r((size_t)V, (float)(1.0 / V)), old((size_t)V);
    for (int it = 0; it < iterations; it++) {
        old = r;
        double dangling = 0;
        for (int v = 0; v < V; v++) if (indeg[v] == 0) dangling += old[v] / V;
        for (int u = 0; u < V; u++) {
            float acc = 0;
            for (long long p = g.rowptr[u]; p < g.rowptr[u + 1]; p++) {
                const int v = g.adj[p];
                if (v != u) acc += old[v] * (float)(1.0 / indeg[v]);
            }
            r[u] = k + d * (acc + (float)dangling);
        }
    }
    return r;
}
inline std::vector<int> seq_components(const HostCSR &g)
{
    std::vector<int> c((size_t)g.V, -1);
    int cur = 1;
    for (int s0 = 0; s0 < g.V; s0++) {
        if (c[s0] != -1) continue;
        std::queue<int> q; c[s0] = cur; q.push(s0);
        while (!q.empty()) {
            const int s = q.front(); q.pop();
            for (long long p = g.rowptr[s]; p < g.rowptr[s + 1]; p++) if (c[g.adj[p]] == -1) { c[g.adj[p]] = cur; q.push(g.adj[p]); }
        }
        cur++;
    }
    return c;
}
// equal_components (verify_results.h:198-254): the two labelings must induce the same partition
inline int equal_components(const std::vector<int> &a, const std::vector<int> &b)
{
    std::map<int, int> f, r; int errors = 0;
    for (size_t i = 0; i < a.size(); i++) {
        auto x = f.emplace(a[i], b[i]); auto y = r.emplace(b[i], a[i]);
        if (x.first->second != b[i] || y.first->second != a[i]) errors++;
    }
    std::cout << "error count: " << errors << std::endl;
    return errors;
}

// sequential Ford-Fulkerson with the rules of apps/algorithms/mf.hpp (seq_mf.hpp:5-154 in spirit): level-synchronous search over
// edges with positive residual, parent = smallest id among the previous level's vertices that reach a vertex; the value of u->v is
// the first match in u's row, an augmentation updates every parallel u->v and every stored v->u
inline int seq_ford_fulkerson(const HostCSR &h, int source, int sink, int capacity)
{
    std::vector<int> flow(h.adj.size(), capacity), parent((size_t)h.V), level((size_t)h.V);
    int total = 0;
    while (source != sink) {
        std::fill(parent.begin(), parent.end(), 0x7FFFFFFF); std::fill(level.begin(), level.end(), -1);
        level[(size_t)source] = 1;
        std::vector<int> front{source}, next;
        for (int cur = 1; !front.empty(); cur++, front.swap(next), next.clear())
            for (int u : front)
                for (long long p = h.rowptr[(size_t)u]; p < h.rowptr[(size_t)u + 1]; p++) {
                    const int v = h.adj[(size_t)p];
                    if (flow[(size_t)p] <= 0 || (level[(size_t)v] != -1 && level[(size_t)v] != cur + 1)) continue;
                    if (level[(size_t)v] == -1) { level[(size_t)v] = cur + 1; next.push_back(v); }
                    parent[(size_t)v] = std::min(parent[(size_t)v], u);
                }
        if (level[(size_t)sink] == -1) break;
        int path_flow = 0x7FFFFFFF;
        for (int v = sink; v != source; v = parent[(size_t)v]) {
            const int u = parent[(size_t)v]; int w = 0;
            for (long long p = h.rowptr[(size_t)u]; p < h.rowptr[(size_t)u + 1]; p++) if (h.adj[(size_t)p] == v) { w = flow[(size_t)p]; break; }
            path_flow = std::min(path_flow, w);
        }
        if (path_flow <= 0) break;
        for (int v = sink; v != source; v = parent[(size_t)v]) {
            const int u = parent[(size_t)v];
            for (long long p = h.rowptr[(size_t)u]; p < h.rowptr[(size_t)u + 1]; p++) if (h.adj[(size_t)p] == v) flow[(size_t)p] -= path_flow;
            for (long long p = h.rowptr[(size_t)v]; p < h.rowptr[(size_t)v + 1]; p++) if (h.adj[(size_t)p] == u) flow[(size_t)p] += path_flow;
        }
        total += path_flow;
    }
    return total;
}
