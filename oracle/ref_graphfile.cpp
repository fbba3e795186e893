/*
 * ref_graphfile.cpp -- ORACLE TOOLING (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Drives the GENUINE reference graph-file code (VGL_Graph::save_to_binary_file / load_from_binary_file,
 * vgl_graph.hpp:109-161; csr_graph.hpp:73-104; vect_csr_graph.hpp:141-181) so that the `.csr` / `.vcsr` fixtures under
 * tests/golden/ are written by the reference itself, and so that files written by this repository can be fed back to it.
 * Includes the reference's umbrella header from /root/reference where it lies; nothing is copied.  Built only by
 * `make -C oracle ref` into oracle/_ref/ (git-ignored).
 *
 *   ref_graphfile save <graph.el_container> <csr|vcsr> <out_file>         import the edge list, write the graph file
 *   ref_graphfile bfs  <graph_file> <csr|vcsr> <source_original_id> <out> load the graph file, run the reference BFS,
 *                                                                         int32 levels[V] in ORIGINAL numbering
 */
#define INT_ELEMENTS_PER_EDGE 4.0
#define NEC_VECTOR_ENGINE_THRESHOLD_VALUE  VECTOR_LENGTH * MAX_SX_AURORA_THREADS * 128
#define VECTOR_CORE_THRESHOLD_VALUE 2*VECTOR_LENGTH
#define COLLECTIVE_FRONTIER_TYPE_CHANGE_THRESHOLD 0.35
#include "graph_library.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s save <el_container> <csr|vcsr> <out> | bfs <graph_file> <csr|vcsr> <source> <out>\n", argv[0]); return 2; }
    try
    {
        VGL_RUNTIME::init_library(argc, argv);
        GraphStorageFormat fmt = (strcmp(argv[3], "vcsr") == 0) ? VECTOR_CSR_GRAPH : CSR_GRAPH;
        if (strcmp(argv[1], "save") == 0)
        {
            EdgesContainer ec;
            if (!ec.load_from_binary_file(argv[2])) { fprintf(stderr, "cannot read %s\n", argv[2]); return 3; }
            VGL_Graph graph(fmt);
            graph.import(ec);
            if (!graph.save_to_binary_file(argv[4])) { fprintf(stderr, "cannot write %s\n", argv[4]); return 3; }
        }
        else if (strcmp(argv[1], "bfs") == 0 && argc >= 6)
        {
            VGL_Graph graph(fmt);
            if (!graph.load_from_binary_file(argv[2])) { fprintf(stderr, "cannot read %s\n", argv[2]); return 3; }
            int source = graph.reorder(atoi(argv[4]), ORIGINAL, SCATTER);
            VerticesArray<int> levels(graph, SCATTER);
            BFS::vgl_top_down(graph, levels, source);
            levels.reorder(ORIGINAL);
            FILE *out = fopen(argv[5], "wb");
            if (!out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 3; }
            fwrite(levels.get_ptr(), sizeof(int), levels.size(), out);
            fclose(out);
        }
        else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
        VGL_RUNTIME::finalize_library();
    }
    catch (string error) { cout << error << endl; return 1; }
    catch (const char *error) { cout << error << endl; return 1; }
    return 0;
}
