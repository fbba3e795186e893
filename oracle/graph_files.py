"""VGL graph files (.csr / .vcsr) -- numpy restatement of the reference's containers and their on-disk layout.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (imported by tests/ and by oracle/make_golden_graph_files.py only).

File layout (VGL_Graph::save_to_binary_file, vgl_graph.hpp:109-130):
    int V; long long E; int container_type;  <outgoing container>  <incoming container>
container (CSRGraph::save_main_content_to_binary_file csr_graph.hpp:73-85,
           VectorCSRGraph::save_main_content_to_binary_file vect_csr_graph.hpp:141-155):
    int V; long long E; int format; long long rowptr[V+1]; int adj[E];
    [VECTOR_CSR_GRAPH only: int forward_conversion[V]; int backward_conversion[V];]  long long edges_reorder_indexes[E]
container_type / format: VECTOR_CSR_GRAPH = 1, CSR_GRAPH = 3 (framework_types.h:49-57).

Containers (VGL_Graph::import, vgl_graph.hpp:57-68): the outgoing one is built from the edge list, which the build leaves
SORTED in out-CSR order; the incoming one is built from that sorted list transposed.  CSR_GRAPH: stable sort by source
(csr/import.hpp:5-58, sorter.h:55-93).  VECTOR_CSR_GRAPH: every direction first renumbers the vertices by its own degree,
largest first, stable (vect_csr/import.hpp:61-99,257-337), then sorts the renumbered list by source.
Pinned by tests/golden/*.csr / *.vcsr, which the reference wrote itself (oracle/ref_graphfile.cpp).
"""
import numpy as np

VECTOR_CSR_GRAPH = 1
CSR_GRAPH = 3


def _direction(V, src, dst, fmt):
    """one container from the current edge list; returns the container and the list as the build leaves it"""
    src = np.asarray(src, np.int32)
    dst = np.asarray(dst, np.int32)
    c = {}
    if fmt == VECTOR_CSR_GRAPH:
        deg = np.bincount(src, minlength=V)
        bwd = np.argsort(-deg, kind="stable").astype(np.int32)        # stored id -> original id
        fwd = np.empty(V, np.int32)
        fwd[bwd] = np.arange(V, dtype=np.int32)                         # original id -> stored id
        c["fwd"], c["bwd"] = fwd, bwd
        key, val = fwd[src], fwd[dst]
    else:
        key, val = src, dst
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(V + 1, np.int64)
    np.cumsum(np.bincount(key, minlength=V), out=rowptr[1:])
    c["rowptr"], c["adj"], c["perm"] = rowptr, val[order].astype(np.int32), order.astype(np.int64)
    return c, src[order], dst[order]


def build_graph(V, src, dst, fmt):
    """{'format', 'V', 'E', 'out': container, 'in': container} as VGL_Graph::import builds them"""
    out, s1, d1 = _direction(V, src, dst, fmt)
    inc, _, _ = _direction(V, d1, s1, fmt)
    return {"format": fmt, "V": int(V), "E": int(len(src)), "out": out, "in": inc}


def graph_file_bytes(g):
    head = np.int32(g["V"]).tobytes() + np.int64(g["E"]).tobytes() + np.int32(g["format"]).tobytes()
    parts = [head]
    for d in ("out", "in"):
        c = g[d]
        parts += [head, c["rowptr"].astype(np.int64).tobytes(), c["adj"].astype(np.int32).tobytes()]
        if g["format"] == VECTOR_CSR_GRAPH:
            parts += [c["fwd"].astype(np.int32).tobytes(), c["bwd"].astype(np.int32).tobytes()]
        parts.append(c["perm"].astype(np.int64).tobytes())
    return b"".join(parts)


def write_graph_file(path, V, src, dst, fmt):
    with open(path, "wb") as f:
        f.write(graph_file_bytes(build_graph(V, src, dst, fmt)))


def read_graph_file(path):
    raw = open(path, "rb").read()
    pos = 0

    def take(dtype, n):
        nonlocal pos
        a = np.frombuffer(raw, dtype, n, pos)
        pos += a.nbytes
        return a

    V, E, fmt = int(take(np.int32, 1)[0]), int(take(np.int64, 1)[0]), int(take(np.int32, 1)[0])
    if fmt not in (CSR_GRAPH, VECTOR_CSR_GRAPH):
        raise ValueError("unsupported container type %d" % fmt)
    g = {"format": fmt, "V": V, "E": E}
    for d in ("out", "in"):
        cv, ce, cf = int(take(np.int32, 1)[0]), int(take(np.int64, 1)[0]), int(take(np.int32, 1)[0])
        if (cv, ce, cf) != (V, E, fmt):
            raise ValueError("container header does not match the file header")
        c = {"rowptr": take(np.int64, V + 1), "adj": take(np.int32, E)}
        if fmt == VECTOR_CSR_GRAPH:
            c["fwd"], c["bwd"] = take(np.int32, V), take(np.int32, V)
        c["perm"] = take(np.int64, E)
        g[d] = c
    if pos != len(raw):
        raise ValueError("trailing bytes in graph file")
    return g


def edges_in_original_ids(g, direction="out"):
    """(src, dst) of the stored direction, ORIGINAL vertex ids, in CSR order (for the incoming container: src = row)"""
    c = g[direction]
    rows = np.repeat(np.arange(g["V"], dtype=np.int32), np.diff(c["rowptr"]))
    if g["format"] == VECTOR_CSR_GRAPH:
        return c["bwd"][rows], c["bwd"][c["adj"]]
    return rows, c["adj"]
