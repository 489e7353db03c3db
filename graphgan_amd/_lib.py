"""ctypes binding of ``libgraphgan_hip.so`` (C ABI: ``include/graphgan_hip.h``).

The library is built in-tree by ``make -C graphgan_amd/csrc`` (or
``__graft_entry__.build()``).  There is no fallback: if the shared object is
missing this module raises at import, and without a gfx950 device
``gg_create`` fails with ``GG_EHIP``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgraphgan_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "graphgan_hip.h")
ABI_VERSION = 7  # == GG_ABI_VERSION of include/graphgan_hip.h (tests/test_host_cpu.py keeps header, binding and library in step)


def header_abi_version(path=HEADER_PATH):
    """GG_ABI_VERSION as the C header declares it (the one place the number is defined)."""
    import re
    with open(path) as f:
        m = re.search(r"^#define\s+GG_ABI_VERSION\s+(\d+)", f.read(), flags=re.M)
    if not m:
        raise RuntimeError("GG_ABI_VERSION not found in %s" % path)
    return int(m.group(1))

GG_OK, GG_EINVAL, GG_ECAPACITY, GG_EHIP, GG_ECOMM, GG_ENOMEM, GG_EIO = 0, -1, -2, -3, -4, -5, -6
GG_OPT_ADAM_DENSE, GG_OPT_ADAM_LAZY, GG_OPT_SGD = 0, 1, 2
GG_ROOT_OK, GG_ROOT_ABORTED, GG_ROOT_EMPTY = 0, 1, 2
ERROR_NAMES = {GG_EINVAL: "GG_EINVAL", GG_ECAPACITY: "GG_ECAPACITY", GG_EHIP: "GG_EHIP", GG_ECOMM: "GG_ECOMM",
               GG_ENOMEM: "GG_ENOMEM", GG_EIO: "GG_EIO"}


class GGConfig(ctypes.Structure):
    _fields_ = [("lr_gen", ctypes.c_float), ("lr_dis", ctypes.c_float),
                ("lambda_gen", ctypes.c_float), ("lambda_dis", ctypes.c_float),
                ("adam_beta1", ctypes.c_float), ("adam_beta2", ctypes.c_float), ("adam_eps", ctypes.c_float),
                ("window_size", ctypes.c_int32), ("optimizer", ctypes.c_int32), ("device", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 6)]


class GGCounters(ctypes.Structure):
    _fields_ = [("walks", ctypes.c_int64), ("hops", ctypes.c_int64), ("nbr_reads", ctypes.c_int64),
                ("reward_pairs", ctypes.c_int64), ("d_pairs", ctypes.c_int64), ("g_pairs", ctypes.c_int64),
                ("d_steps", ctypes.c_int64), ("g_steps", ctypes.c_int64),
                ("last_kernel_ms", ctypes.c_double), ("walk_kernel_ms", ctypes.c_double),
                ("walk_launches", ctypes.c_int64), ("rows_scored", ctypes.c_int64), ("score_kernel_ms", ctypes.c_double),
                ("score_launches", ctypes.c_int64), ("score_chunks", ctypes.c_int64), ("score_rows", ctypes.c_int64),
                ("bfs_kernel_ms", ctypes.c_double), ("bfs_trees", ctypes.c_int64), ("score_dists", ctypes.c_int64),
                ("reward_kernel_ms", ctypes.c_double), ("reward_pairs_timed", ctypes.c_int64),
                ("d_grad_ms", ctypes.c_double), ("d_opt_ms", ctypes.c_double), ("d_pairs_timed", ctypes.c_int64), ("d_rows_timed", ctypes.c_int64),
                ("g_grad_ms", ctypes.c_double), ("g_opt_ms", ctypes.c_double), ("g_pairs_timed", ctypes.c_int64), ("g_rows_timed", ctypes.c_int64),
                ("d_passes_timed", ctypes.c_int64), ("g_passes_timed", ctypes.c_int64),
                ("g_walk_nodes_timed", ctypes.c_int64),
                ("es_gathers", ctypes.c_int64), ("es_nodes", ctypes.c_int64), ("score_gathers", ctypes.c_int64), ("score_nodes", ctypes.c_int64),
                ("walk_reruns", ctypes.c_int64)]


class GGGraph(ctypes.Structure):
    _fields_ = [("n_node", ctypes.c_int32), ("nnz", ctypes.c_int64), ("n_train_edges", ctypes.c_int64),
                ("n_test_edges", ctypes.c_int64), ("rowptr", ctypes.POINTER(ctypes.c_int64)), ("col", ctypes.POINTER(ctypes.c_int32))]


class GraphGANHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "GG_E?"), code, msg))
        self.code = code


_P = ctypes.c_void_p
_i32, _i64, _u32, _u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64

# name -> (restype, argtypes); exactly the declarations of include/graphgan_hip.h
SIGNATURES = {
    "gg_abi_version": (ctypes.c_int, []),
    "gg_last_error": (ctypes.c_char_p, [_P]),
    "gg_create": (ctypes.c_int, [_i32, _i32, _P, _P, ctypes.POINTER(GGConfig), ctypes.POINTER(_P)]),
    "gg_destroy": (ctypes.c_int, [_P]),
    "gg_set_graph_csr": (ctypes.c_int, [_P, _P, _P]),
    "gg_host_build_trees": (_i64, [_i32, _P, _P, _P, _i32, _P, _P, _P, _i64, _i32, _P]),
    "gg_build_trees": (ctypes.c_int, [_P, _P, _i32, _i32]),
    "gg_build_trees_device": (ctypes.c_int, [_P, _P, _i32]),
    "gg_set_tree_mode": (ctypes.c_int, [_P, _i32, _i64]),
    "gg_debug_words": (ctypes.c_int, [_P, _i32, _i32, _P]),
    "gg_lazy_stats": (ctypes.c_int, [_P, _P]),
    "gg_get_lazy_trees": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "gg_set_trees": (ctypes.c_int, [_P, _P, _i32, _P, _P, _P, _i32]),
    "gg_tree_info": (ctypes.c_int, [_P, _P, _P, _P]),
    "gg_tree_roots": (ctypes.c_int, [_P, _P]),
    "gg_save_trees": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "gg_load_trees": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "gg_get_trees": (ctypes.c_int, [_P, _P, _P, _P]),
    "gg_get_tree_order": (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    "gg_walk_sample": (ctypes.c_int, [_P, _P, _P, _i32, _i32, _u64, _u32, _P, _P, _P, _i32, _P]),
    "gg_walk_info": (ctypes.c_int, [_P, _P, _P, _P]),
    "gg_get_walks": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "gg_prepare_d": (ctypes.c_int, [_P, _P, _i32, _u64, _u32, _P, _P]),
    "gg_get_d_data": (ctypes.c_int, [_P, _P, _P, _P]),
    "gg_prepare_g": (ctypes.c_int, [_P, _P, _i32, _i32, _u64, _u32, _P, _P]),
    "gg_prepare_g_begin": (ctypes.c_int, [_P, _P, _i32, _i32, _u64, _u32]),
    "gg_get_g_data": (ctypes.c_int, [_P, _P, _P, _P]),
    "gg_epoch_begin": (ctypes.c_int, [_P, _i32, _i32]),
    "gg_epoch_add": (ctypes.c_int, [_P, _P, _i32, _i32, _i32, _i32, _u64, _u32, _u32, _P, _P]),
    "gg_epoch_commit": (ctypes.c_int, [_P, _i32, _P]),
    "gg_q3_clear": (ctypes.c_int, [_P]),
    "gg_q3_get": (ctypes.c_int, [_P, _P, _P]),
    "gg_d_pass": (ctypes.c_int, [_P, _P, _i64, _i32]),
    "gg_g_pass": (ctypes.c_int, [_P, _P, _i64, _i32]),
    "gg_pair_reward": (ctypes.c_int, [_P, _P, _P, _i64, _P]),
    "gg_d_step": (ctypes.c_int, [_P, _P, _P, _P, _i32]),
    "gg_g_step": (ctypes.c_int, [_P, _P, _P, _P, _i32]),
    "gg_all_score": (ctypes.c_int, [_P, _P, _i32, _P]),
    "gg_all_score_reduce": (ctypes.c_int, [_P, _P, _i32, _i32, _i32, _P, _P, _P, _P]),
    "gg_get_embeddings": (ctypes.c_int, [_P, _i32, _P]),
    "gg_get_bias": (ctypes.c_int, [_P, _i32, _P]),
    "gg_write_embeddings": (ctypes.c_int, [_P, _i32, ctypes.c_char_p, _i32]),
    "gg_host_write_embeddings": (ctypes.c_int, [_P, _i64, _i32, ctypes.c_char_p, _i32]),
    "gg_write_embeddings_bin": (ctypes.c_int, [_P, _i32, ctypes.c_char_p]),
    "gg_edge_scores": (ctypes.c_int, [_P, _i32, _P, _P, _i64, _P]),
    "gg_set_embeddings": (ctypes.c_int, [_P, _i32, _P]),
    "gg_set_bias": (ctypes.c_int, [_P, _i32, _P]),
    "gg_save_state": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "gg_load_state": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "gg_get_counters": (ctypes.c_int, [_P, ctypes.POINTER(GGCounters)]),
    "gg_set_profiling": (ctypes.c_int, [_P, _i32]),
    "gg_set_profiling_solo": (ctypes.c_int, [_P, _i32]),
    "gg_synchronize": (ctypes.c_int, [_P]),
    "gg_comm_unique_id": (ctypes.c_int, [_P]),
    "gg_comm_init": (ctypes.c_int, [_P, _P, _i32, _i32]),
    "gg_comm_barrier": (ctypes.c_int, [_P]),
    "gg_comm_stats": (ctypes.c_int, [_P, _P]),
    "gg_comm_stats_ex": (ctypes.c_int, [_P, _P]),
    "gg_synth_powerlaw": (_i64, [_i32, _i32, _u64, _u64, _P, _i64]),
    "gg_host_read_edges": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(GGGraph)]),
    "gg_host_free_graph": (None, [ctypes.POINTER(GGGraph)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "graphgan_amd: %s is missing -- build the HIP extension first "
            "(`make -C graphgan_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gg_abi_version() != ABI_VERSION:
        raise ImportError("graphgan_amd: %s has ABI version %d, this binding expects %d -- rebuild (make -C graphgan_amd/csrc)"
                          % (LIB_PATH, lib.gg_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, ctx=None):
    if rc is not None and rc < 0:
        msg = lib.gg_last_error(ctx)
        raise GraphGANHipError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
