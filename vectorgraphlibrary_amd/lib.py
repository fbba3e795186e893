"""ctypes binding of libvgl_hip.so (the C ABI declared in include/vgl_hip.h).

There is no CPU fallback: if the shared object is missing or no HIP device is visible the
package raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C vectorgraphlibrary_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvgl_hip.so")


class VglHipError(RuntimeError):
    pass


class BfsStats(C.Structure):
    _fields_ = [("levels", C.c_int32), ("td_steps", C.c_int32), ("bu_steps", C.c_int32),
                ("edges_examined", C.c_int64), ("frontier_total", C.c_int64), ("discovered", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("td_edges", C.c_int64), ("td_frontier", C.c_int64),
                ("bu_edges", C.c_int64), ("bu_found", C.c_int64)]


class SsspStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("edges_relaxed", C.c_int64), ("algorithmic_bytes", C.c_int64),
                ("push_steps", C.c_int32), ("pull_steps", C.c_int32)]


class PrStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("ranks_sum", C.c_double), ("algorithmic_bytes", C.c_int64)]


class SccStats(C.Structure):
    _fields_ = [("trim_rounds", C.c_int32), ("forward_backward_steps", C.c_int32), ("colour_rounds", C.c_int32), ("edge_passes", C.c_int32)]


class CcStats(C.Structure):
    _fields_ = [("hook_passes", C.c_int32), ("algorithmic_bytes", C.c_int64)]


class ExchangeStats(C.Structure):
    _fields_ = [("collectives", C.c_int64), ("bytes_received", C.c_int64), ("list_steps", C.c_int32), ("dense_steps", C.c_int32),
                ("sparse_levels", C.c_int32), ("exchanges", C.c_int32)]


_lib = None

_p, _i32, _i64, _u64, _int, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_int, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (argtypes)  ; every entry returns int status unless listed in _SPECIAL
_SIGNATURES = {
    "vgl_hip_ctx_create": [_int, _p, _pp],
    "vgl_hip_ctx_destroy": [_p],
    "vgl_hip_ctx_sync": [_p],
    "vgl_hip_ctx_trim": [_p],
    "vgl_hip_malloc": [_p, C.c_size_t, _pp],
    "vgl_hip_free": [_p, _p],
    "vgl_hip_memcpy_h2d": [_p, _p, _p, C.c_size_t],
    "vgl_hip_memcpy_d2h": [_p, _p, _p, C.c_size_t],
    "vgl_hip_memset": [_p, _p, _int, C.c_size_t],
    "vgl_hip_gen_rmat": [_p, _int, _i64, _i64, _u64, _int, _int, _int, _int, _int, _p, _p],
    "vgl_hip_gen_uniform": [_p, _int, _i64, _i64, _u64, _p, _p],
    "vgl_hip_gen_weights": [_p, _i64, _i64, _u64, _p],
    "vgl_hip_coo_to_csr": [_p, _i32, _i64, _p, _p, _i32, _i32, _p, _p, _p, C.POINTER(_i64)],
    "vgl_hip_gather_u32": [_p, _i64, _p, _p, _p],
    "vgl_hip_degree_order": [_p, _i32, _i64, _p, _p, _int, _p, _p],
    "vgl_hip_degree_hist_add": [_p, _i64, _p, _p, _int, _p],
    "vgl_hip_degree_order_from_degrees": [_p, _i32, _p, _p, _p],
    "vgl_hip_relabel_i32": [_p, _i64, _p, _p, _p],
    "vgl_hip_permute_u32": [_p, _i64, _p, _p, _p],
    "vgl_hip_cc_labels_to_original": [_p, _i32, _p, _p, _p, _p, _p],
    "vgl_hip_partition_rows": [_p, _i32, _p, _int, C.POINTER(_i32)],
    "vgl_hip_graph_create": [_p, _i32, _i32, _i32, _p, _p, _i64, _p, _p, _i64, _pp],
    "vgl_hip_graph_destroy": [_p, _p],
    "vgl_hip_frontier_create": [_p, _p, _pp],
    "vgl_hip_frontier_destroy": [_p, _p],
    "vgl_hip_frontier_create_on": [_p, _p, _p, _p, _pp],
    "vgl_hip_frontier_set_state": [_p, _p, _p, _i32, _i64, _int],
    "vgl_hip_frontier_set_all_active": [_p, _p],
    "vgl_hip_frontier_clear": [_p, _p],
    "vgl_hip_frontier_add_vertex": [_p, _p, _i32],
    "vgl_hip_frontier_info": [_p, _p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_int)],
    "vgl_hip_gnf_from_flags": [_p, _p, _p, _dbl, _p],
    "vgl_hip_gnf_equal_i32": [_p, _p, _p, _i32, _dbl, _p],
    "vgl_hip_graph_tile_rows": [_p, _int, _pp, C.POINTER(_i64)],
    "vgl_hip_frontier_advance_plan": [_p, _p, _p, _int, _pp, _pp, C.POINTER(_i64)],
    "vgl_hip_reduce_sum_f64_buffer": [_p, _i64, _p, C.POINTER(_dbl)],
    "vgl_hip_gnf_begin": [_p, _p, _p, _int, _p],
    "vgl_hip_gnf_complete": [_p, _p, _p, _dbl, _int, _i64],
    "vgl_hip_reduce_sum_i32": [_p, _p, _p, C.POINTER(_i64)],
    "vgl_hip_reduce_sum_f32": [_p, _p, _p, C.POINTER(_dbl)],
    "vgl_hip_count_not_equal_u32": [_p, _i32, _p, _p, C.POINTER(_i64)],
    "vgl_hip_bfs_run": [_p, _p, _i32, _int, _p, C.POINTER(BfsStats)],
    "vgl_hip_bfs_run_batch": [_p, _p, C.POINTER(_i32), _i32, _int, _p, C.POINTER(BfsStats)],
    "vgl_hip_bfs_prepare_blocked": [_p, _p],
    "vgl_hip_sssp_run": [_p, _p, _p, _i32, _int, _p, C.POINTER(SsspStats)],
    "vgl_hip_sswp_run": [_p, _p, _p, _i32, _int, _p, C.POINTER(SsspStats)],
    "vgl_hip_sssp_run_delta": [_p, _p, _p, _i32, C.c_float, _p, C.POINTER(SsspStats)],
    "vgl_hip_sssp_pull_plan_create": [_p, _p, _p, _pp],
    "vgl_hip_sssp_pull_plan_destroy": [_p, _p],
    "vgl_hip_sssp_pull_pass": [_p, _p, _p, _p, C.POINTER(_int)],
    "vgl_hip_sssp_pull_plan_info": [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_sssp_run_pull": [_p, _p, _p, _p, _i32, _int, _p, C.POINTER(SsspStats)],
    "vgl_hip_sswp_run_pull": [_p, _p, _p, _p, _i32, _int, _p, C.POINTER(SsspStats)],
    "vgl_hip_sssp_plan_create": [_p, _p, _p, C.c_float, _pp],
    "vgl_hip_sssp_plan_destroy": [_p, _p],
    "vgl_hip_sssp_run_plan": [_p, _p, _p, _i32, _p, C.POINTER(SsspStats)],
    "vgl_hip_pr_run": [_p, _p, _p, _int, _p, C.POINTER(PrStats)],
    "vgl_hip_pr_run_mode": [_p, _p, _p, _int, _int, _p, C.POINTER(PrStats)],
    "vgl_hip_pr_prepare": [_p, _p, _int, C.POINTER(_int)],
    "vgl_hip_cc_prepare": [_p, _p],
    "vgl_hip_hits_run": [_p, _p, _int, _p, _p],
    "vgl_hip_scc_run": [_p, _p, _p, C.POINTER(SccStats)],
    "vgl_hip_cc_run": [_p, _p, _p, C.POINTER(CcStats)],
    "vgl_hip_cc_run_symmetric": [_p, _p, _p, C.POINTER(CcStats)],
    "vgl_hip_bfs_init": [_p, _i32, _i32, _p],
    "vgl_hip_bfs_step_top_down": [_p, _p, _p, _i32, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_bfs_step_top_down_bits": [_p, _p, _p, _i32, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_bitmap_or_parts": [_p, _i64, _int, _p, _p],
    "vgl_hip_bfs_step_bottom_up": [_p, _p, _p, _i32, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_levels_to_bitmap": [_p, _i32, _p, _i32, _p],
    "vgl_hip_bfs_apply_bitmaps": [_p, _i32, _int, _p, _p, _i32, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_bfs_apply_bitmaps_owned": [_p, _i32, _int, _p, _p, _i32, _p, _p, _p, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_bitmap_to_ids": [_p, _i64, _p, _i32, _p],
    "vgl_hip_bfs_apply_ids": [_p, _i32, _int, _i32, _p, _p, _i32, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)],
    "vgl_hip_sssp_init": [_p, _i32, _i32, _p],
    "vgl_hip_sssp_relax_owned": [_p, _p, _p, _p, C.POINTER(_int)],
    "vgl_hip_sswp_init": [_p, _i32, _i32, _p],
    "vgl_hip_sswp_relax_owned": [_p, _p, _p, _p, C.POINTER(_int)],
    "vgl_hip_cc_init": [_p, _i32, _p],
    "vgl_hip_cc_hook_owned": [_p, _p, _p, C.POINTER(_int)],
    "vgl_hip_cc_jump": [_p, _i32, _p],
    "vgl_hip_pr_setup": [_p, _i32, _p, _p, _p],
    "vgl_hip_sum_over_edges_f32": [_p, _p, _p, C.c_float, _p],
    "vgl_hip_sssp_prepare": [_p, _p],
    "vgl_hip_comm_set_timeout_ms": [_p, C.c_double],
    "vgl_hip_pr_iteration_owned": [_p, _p, _p, _p, _p, _p],
    "vgl_hip_indegree_noloops_add": [_p, _p, _p],
    "vgl_hip_diff_to_pairs_u32": [_p, _i32, _p, _p, _i32, _p],
    "vgl_hip_apply_pairs_u32": [_p, _int, _i64, _int, _p, _int, _i32, _p, C.POINTER(_int)],
    "vgl_hip_comm_unique_id": [_p],
    "vgl_hip_comm_create": [_p, _int, _int, _p, _pp],
    "vgl_hip_comm_create_hosted": [_p, _int, _int, C.c_char_p, C.c_size_t, _pp],
    "vgl_hip_comm_create_peer": [_p, _int, _int, C.c_char_p, C.c_size_t, _pp],
    "vgl_hip_comm_abort": [_p],
    "vgl_hip_comm_destroy": [_p],
    "vgl_hip_comm_info": [_p, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)],
    "vgl_hip_comm_barrier": [_p],
    "vgl_hip_comm_stats": [_p, C.POINTER(ExchangeStats)],
    "vgl_hip_exchange_allreduce_min_i32": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_min_f32": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_max_f32": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_sum_i32": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_sum_i64": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_sum_f32": [_p, _p, _i64],
    "vgl_hip_exchange_allreduce_sum_f64": [_p, _p, _i64],
    "vgl_hip_exchange_allgather": [_p, _p, _p, _i64],
    "vgl_hip_exchange_allgather_slices": [_p, _p, C.POINTER(_i64), _int],
    "vgl_hip_exchange_bitmap_or": [_p, _p, _i64],
    "vgl_hip_exchange_changed_u32": [_p, _i32, _p, _p, _int, C.POINTER(_int)],
    "vgl_hip_bfs_run_sharded": [_p, _p, _p, _i32, _int, _i64, _int, _p, C.POINTER(BfsStats)],
    "vgl_hip_sssp_run_sharded": [_p, _p, _p, _p, _i32, _p, C.POINTER(SsspStats)],
    "vgl_hip_sswp_run_sharded": [_p, _p, _p, _p, _i32, _p, C.POINTER(SsspStats)],
    "vgl_hip_cc_run_sharded": [_p, _p, _p, _p, C.POINTER(CcStats)],
    "vgl_hip_pr_run_sharded": [_p, _p, _p, _int, _int, _p, C.POINTER(PrStats)],
    "vgl_hip_hits_run_sharded": [_p, _p, _p, _int, _p, _p],
    "vgl_hip_timing_enable": [_p, _int],
    "vgl_hip_timing_only": [_p, C.c_char_p],
    "vgl_hip_timing_reset": [_p],
    "vgl_hip_timing_stride": [_p, _int],
    "vgl_hip_timing_get": [_p, C.c_char_p, C.POINTER(_i64), C.POINTER(_dbl)],
}
_SPECIAL = {
    "vgl_hip_abi_version": (_int, []),
    "vgl_hip_last_error": (C.c_char_p, []),
    "vgl_hip_ctx_stream": (_p, [_p]),
    "vgl_hip_frontier_ids": (_p, [_p]),
    "vgl_hip_frontier_flags": (_p, [_p]),
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_SPECIAL))


def load():
    """dlopen libvgl_hip.so and declare every entry point; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VglHipError(
            f"{LIB_PATH} is missing: the MI355X backend has no CPU fallback. "
            "Build it with `make -C vectorgraphlibrary_amd/csrc` (hipcc, gfx950).")
    L = C.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = _int
        fn.argtypes = args
    for name, (res, args) in _SPECIAL.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status):
    if status != 0:
        raise VglHipError(load().vgl_hip_last_error().decode("utf-8", "replace"))
