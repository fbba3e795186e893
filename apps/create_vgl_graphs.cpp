// create_vgl_graphs: counterpart of apps/utilites/create_vgl_graphs.cpp:7-75 -- generate (or take an .el_container with -import) an
// edge list and save it in one of the reference's file formats:  -format el_container | csr | vcsr,  -file <name without extension>.
// -load <graph file> re-saves a graph file instead (load + save round trip).
// The files are the reference's own layouts (edges_container.h:58-99, vgl_graph.hpp:109-130): either side can read the other's.
#include "common.hpp"
int main(int argc, char **argv)
{
    try {
        VGL_RUNTIME::init_library(argc, argv);
        Parser parser;
        std::string out_name = "graph", in_name, graph_in_name;
        {   // -file names the OUTPUT here (as in the reference's tool); -import names an input edge list
            std::vector<char *> rest{argv[0]};
            for (int i = 1; i < argc; i++) {
                const std::string a = argv[i];
                if ((a == "-file" || a == "-f") && i + 1 < argc) out_name = argv[++i];
                else if (a == "-import" && i + 1 < argc) in_name = argv[++i];
                else if (a == "-load" && i + 1 < argc) graph_in_name = argv[++i];      // extension of this tool: re-save a graph file
                else rest.push_back(argv[i]);
            }
            parser.parse_args((int)rest.size(), rest.data());
        }
        GraphGenerationAPI::seed() = parser.seed;
        if (!graph_in_name.empty()) {
            VGL_Graph g(parser.format);
            if (!g.load_from_binary_file(graph_in_name)) throw "Error: graph file not found";
            const std::string name = add_extension(out_name, g.get_format());
            if (!g.save_to_binary_file(name)) throw "Error: can not write the graph file";
            std::cout << "saved " << name << std::endl;
            VGL_RUNTIME::finalize_library();
            return 0;
        }
        EdgesContainer edges_container;
        Timer tm;
        tm.start();
        if (!in_name.empty()) {
            if (!edges_container.load_from_binary_file(in_name)) throw "Error: edges container file not found";
        } else {
            const int v = 1 << parser.scale;
            const DirectionType dir = parser.undirected ? UNDIRECTED_GRAPH : DIRECTED_GRAPH;        // -undirected stores every edge both ways
            if (parser.rmat) GraphGenerationAPI::R_MAT(edges_container, v, (long long)v * parser.avg_degree, 57, 19, 19, 5, dir);
            else GraphGenerationAPI::random_uniform(edges_container, v, (long long)v * parser.avg_degree, dir);
        }
        tm.end();
        tm.print_time_stats(in_name.empty() ? "Generate" : "Load edges container");
        const std::string full_name = add_extension(out_name, parser.format);
        tm.start();
        if (parser.format == EDGES_CONTAINER) {
            if (!edges_container.save_to_binary_file(full_name)) throw "Error: can not write the edges container file";
        } else if (parser.format == VECTOR_CSR_GRAPH) {
            // both per-direction containers of the file on the device (vgl_write_vect_csr_file): the same bytes as import + save_to_binary_file
            if (!vgl_write_vect_csr_file(edges_container, full_name)) throw "Error: can not write the graph file";
        } else {
            VGL_Graph out_graph(parser.format);
            out_graph.import(edges_container);
            if (!out_graph.save_to_binary_file(full_name)) throw "Error: can not write the graph file";
        }
        tm.end();
        tm.print_time_stats("Import and save");
        std::cout << "saved " << full_name << std::endl;
        VGL_RUNTIME::finalize_library();
    } catch (std::string error) { std::cout << error << std::endl; return 1; }
    catch (const char *error) { std::cout << error << std::endl; return 1; }
    return 0;
}
