"""ctypes binding of libmdfri_hip.so (C ABI: include/mdfri.h).

The library is the product: there is no CPU fallback.  Importing this module never touches the GPU;
the first compute call does.  If the shared library has not been built the import of any mDeepFRI
compute module fails loudly with the build command.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDFRI_HIP_LIB", os.path.join(os.path.dirname(_PKG_DIR), "lib", "libmdfri_hip.so"))

MDF_OK, MDF_EINVAL, MDF_ENODEVICE, MDF_ENOMEM, MDF_EBADCHAR, MDF_ECAPACITY, MDF_EIO = 0, -1, -2, -3, -4, -5, -6
MDF_PLAN_KEEP_ORDER = 1      # mdf_plan_create_ex flag: visit the proteins as given, not shortest first
DT_I32, DT_F32, DT_I64, DT_F64, DT_U8 = 0, 1, 2, 3, 4


class MdfriError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmdfri_hip error {code}: {message}")
        self.code = code


class CapacityError(MdfriError):
    pass


class GcnWeights(ctypes.Structure):
    _fields_ = [
        ("embed", c_int32), ("n_gc", c_int32), ("gc_dims", c_int32 * 3), ("fc_dim", c_int32), ("n_terms", c_int32),
        ("W_aa", POINTER(c_float)), ("W_gc", POINTER(c_float) * 3), ("W_fc", POINTER(c_float)),
        ("b_fc", POINTER(c_float)), ("W_out", POINTER(c_float)), ("b_out", POINTER(c_float)),
        ("lm_dim", c_int32), ("W_lm", POINTER(c_float)), ("b_lm", POINTER(c_float)),
        ("embed_linear", c_int32),
    ]


class LmWeights(ctypes.Structure):
    _fields_ = [
        ("hidden", c_int32),
        ("W1", POINTER(c_float)), ("U1", POINTER(c_float)), ("b1", POINTER(c_float)),
        ("W2", POINTER(c_float)), ("U2", POINTER(c_float)), ("b2", POINTER(c_float)),
    ]


class CnnWeights(ctypes.Structure):
    _fields_ = [
        ("n_branch", c_int32), ("kernel_len", POINTER(c_int32)), ("filters", POINTER(c_int32)), ("pad_left", POINTER(c_int32)),
        ("W", POINTER(POINTER(c_float))), ("b", POINTER(POINTER(c_float))),
        ("bn_gamma", POINTER(c_float)), ("bn_beta", POINTER(c_float)), ("bn_mean", POINTER(c_float)), ("bn_var", POINTER(c_float)),
        ("bn_eps", c_float), ("n_terms", c_int32), ("W_out", POINTER(c_float)), ("b_out", POINTER(c_float)),
    ]


class EngineConfig(ctypes.Structure):
    """mdf_engine_config (include/mdfri.h)"""
    _fields_ = [
        ("max_rows", c_int32), ("nnz_per_row", c_int32), ("threshold", c_double), ("generated_contacts", c_int32),
        ("max_segment_groups", c_int32), ("lm_batch", c_int32), ("lm_workspace_gib", c_double), ("graph_max_chunks", c_int32),
        ("pipeline_contact", c_int32),
    ]


class BatchDev(ctypes.Structure):
    """mdf_batch_dev (include/mdfri.h): device pointers of one uploaded batch"""
    _fields_ = [
        ("B", c_int32), ("seqs", c_void_p), ("seq_off", c_void_p), ("Lq", c_void_p), ("coords", c_void_p), ("coord_off", c_void_p),
        ("q_aln", c_void_p), ("t_aln", c_void_p), ("aln_off", c_void_p), ("status", c_void_p), ("bad", c_void_p),
    ]


_f32p, _i32p, _i64p, _u8p = POINTER(c_float), POINTER(c_int32), POINTER(c_int64), POINTER(c_uint8)

# name -> (restype, argtypes); every symbol include/mdfri.h declares
SIGNATURES = {
    "mdf_last_error": (c_char_p, []),
    "mdf_version": (c_char_p, []),
    "mdf_device_count": (c_int, []),
    "mdf_current_device": (c_int, []),
    "mdf_pairwise_sqeuclidean_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "mdf_threshold_lt_i32": (c_int, [c_void_p, c_int64, c_float, c_void_p]),
    "mdf_threshold_lt_f64_i32": (c_int, [c_void_p, c_int64, c_double, c_void_p]),
    "mdf_argwhere_eq1_i32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, _i64p]),
    "mdf_calculate_contact_map": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p, c_int64, _i64p]),
    "mdf_align_len": (c_int, [c_char_p, c_char_p, c_int64, _i64p]),
    "mdf_align_contact_map": (c_int, [c_char_p, c_char_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int]),
    "mdf_build_align_contact_map": (c_int, [c_void_p, c_int64, c_char_p, c_char_p, c_int64, c_double, c_int, c_void_p]),
    "mdf_seq2onehot": (c_int, [c_char_p, c_int64, c_void_p, _i64p]),
    "mdf_model_create": (c_int, [POINTER(GcnWeights), c_int, POINTER(c_void_p)]),
    "mdf_model_load": (c_int, [c_char_p, c_int, POINTER(c_void_p)]),
    "mdf_model_free": (None, [c_void_p]),
    "mdf_model_num_terms": (c_int, [c_void_p]),
    "mdf_model_feature_dim": (c_int, [c_void_p]),
    "mdf_model_device": (c_int, [c_void_p]),
    "mdf_model_lm_dim": (c_int, [c_void_p]),
    "mdf_model_lm": (c_void_p, [c_void_p]),
    "mdf_lm_create": (c_int, [POINTER(LmWeights), c_int, POINTER(c_void_p)]),
    "mdf_lm_free": (None, [c_void_p]),
    "mdf_lm_hidden": (c_int, [c_void_p]),
    "mdf_model_attach_lm": (c_int, [c_void_p, c_void_p]),
    "mdf_lm_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32]),
    "mdf_lm_forward_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_gcn_embed_lm_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "mdf_cnn_create": (c_int, [POINTER(CnnWeights), c_int, POINTER(c_void_p)]),
    "mdf_cnn_free": (None, [c_void_p]),
    "mdf_cnn_num_terms": (c_int, [c_void_p]),
    "mdf_cnn_channels": (c_int, [c_void_p]),
    "mdf_cnn_forward_host": (c_int, [c_void_p, c_char_p, c_int64, c_void_p, _i64p]),
    "mdf_cnn_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int64]),
    "mdf_cnn_padded_channels": (c_int, [c_void_p]),
    "mdf_cnn_pool_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_cnn_head_dev": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "mdf_cnn_forward_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_gcn_forward_host": (c_int, [c_void_p, c_char_p, c_int64, c_void_p, c_int, c_void_p, _i64p]),
    "mdf_layout_rows": (c_int64, [c_void_p, c_int32, c_void_p]),
    "mdf_group_rows": (c_int, []),
    "mdf_hw_pipe": (c_char_p, []),
    "mdf_layer1_form": (c_char_p, []),
    "mdf_seq_encode_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "mdf_cmap_workspace_bytes": (c_size_t, [c_int32, c_int64, c_int32]),
    "mdf_cmap_csr_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                 c_double, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "mdf_cmap_csr_pairs_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                       c_double, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "mdf_cmap_dense_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                                   c_double, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_dense_to_csr_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_gcn_workspace_bytes": (c_size_t, [c_void_p, c_int64]),
    "mdf_letter_sums_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mdf_gcn_embed_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    # the matrix-pipe aggregation (mdf_agg_desc is passed by pointer: opaque here, the batch engine builds it in C++)
    "mdf_gcn_embed_agg_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_gcn_embed_lm_agg_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    "mdf_agg_class": (c_int, [c_int32, c_int]),
    "mdf_agg_l1_fused": (c_int, [c_int32]),
    "mdf_agg_prepare_dev": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "mdf_agg_tile_row_bytes": (c_int32, [c_int32]),
    "mdf_cmap_ws_view": (c_int, [c_void_p, c_size_t, c_int64, c_int32, POINTER(c_void_p), POINTER(c_int32), POINTER(c_void_p)]),
    "mdf_dense_to_csr_masks_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_gcn_pool_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "mdf_head_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "mdf_gcn_head_dev": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdf_plan_create": (c_int, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "mdf_plan_create_ex": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_uint32, POINTER(c_void_p)]),
    "mdf_plan_order": (POINTER(c_int32), [c_void_p, _i64p]),
    "mdf_default_chunk_rows": (c_int32, []),
    "mdf_plan_free": (None, [c_void_p]),
    "mdf_plan_num_proteins": (c_int32, [c_void_p]),
    "mdf_plan_num_chunks": (c_int32, [c_void_p]),
    "mdf_plan_num_segments": (c_int32, [c_void_p]),
    "mdf_plan_max_chunk_rows": (c_int64, [c_void_p]),
    "mdf_plan_chunks": (c_int, [c_void_p, c_void_p]),
    "mdf_plan_segments": (c_int, [c_void_p, c_void_p]),
    "mdf_plan_chunk_row_off": (POINTER(c_int32), [c_void_p, _i64p]),
    "mdf_plan_grp_off": (POINTER(c_int32), [c_void_p, _i64p]),
    "mdf_engine_create": (c_int, [POINTER(c_void_p), c_int32, c_int, POINTER(EngineConfig), POINTER(c_void_p)]),
    "mdf_engine_free": (None, [c_void_p]),
    "mdf_engine_set_nnz_per_row": (c_int, [c_void_p, c_int32]),
    "mdf_engine_nnz_capacity": (c_int64, [c_void_p]),
    "mdf_engine_forward_alignments": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "mdf_engine_forward_dense": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), POINTER(c_void_p), c_int, POINTER(c_void_p), POINTER(c_void_p),
                                         c_void_p]),
    "mdf_engine_check": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), c_void_p, _i64p]),
    "mdf_engine_lm_features_host": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), c_int32, c_void_p, c_void_p]),
    "mdf_engine_num_lms": (c_int32, [c_void_p]),
    "mdf_engine_graph_stats": (c_int, [c_void_p, _i64p, _i64p]),
    "mdf_engine_last_chunk_nnz": (c_int64, [c_void_p, c_void_p]),
    "mdf_seq_engine_create": (c_int, [POINTER(c_void_p), c_int32, c_int, POINTER(c_void_p)]),
    "mdf_seq_engine_free": (None, [c_void_p]),
    "mdf_seq_engine_forward": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), POINTER(c_void_p), c_void_p]),
    "mdf_seq_engine_check": (c_int, [c_void_p, c_void_p, POINTER(BatchDev), c_void_p, _i64p]),
    "mdf_engine_run_alignments_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               POINTER(c_void_p), _i64p]),
    "mdf_engine_submit_alignments_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _i64p]),
    "mdf_engine_collect_host": (c_int, [c_void_p, c_int64, POINTER(c_void_p), _i64p]),
    "mdf_filter_workspace_bytes": (c_size_t, [c_int32]),
    "mdf_filter_scores_dev": (c_int, [c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "mdf_nw_plan": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "mdf_nw_count_long": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32]),
    "mdf_nw_count_long_align": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32]),
    "mdf_nw_score_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "mdf_nw_align_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdf_nw_orient_pairs": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int, c_int]),
    "mdf_nw_score_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int, c_int, c_void_p]),
    "mdf_nw_align_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int, c_int, c_int, c_char_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdf_nw_workspace_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "mdf_nw_best_hits_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_int,
                                       c_int, c_char_p, c_int64, c_int]),
    "mdf_nw_best_hits_align": (c_int, [c_void_p, c_void_p]),
    "mdf_nw_best_hits_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mdf_nw_best_hits_abandon": (c_int, [c_void_p]),
    "mdf_nw_workspace_free": (None, [c_void_p]),
    "mdf_nw_best_hits_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_int, c_int,
                                      c_char_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p]),
    "mdf_matrix_format_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int, _i64p]),
    "mdf_results_format_host": (c_int, [c_void_p, c_void_p, c_char_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int32, c_int32, c_void_p, c_int64, _i64p, _i64p]),
    "mdf_timing_enable": (c_int, [c_int]),
    "mdf_timing_read": (c_int, [c_char_p, _i64p, POINTER(c_double)]),
    "mdf_timing_reset": (c_int, []),
}

_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  If libmdfri_hip.so pulled in the system runtime first, a later
    `import torch` would bind to it by SONAME and find no usable device (two ROCm versions in one process).  Loading
    torch's copy first -- located without importing torch -- makes both share one runtime, whatever the import order."""
    if os.environ.get("MDFRI_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if not spec or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass  # fall back to the system ROCm runtime


def lib() -> ctypes.CDLL:
    """Load libmdfri_hip.so; raise ImportError with the build recipe when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension has not been built and mDeepFRI (MI355X) has no CPU "
                f"fallback.  Build it with `make -C {os.path.join(os.path.dirname(_PKG_DIR), 'csrc')}` "
                f"(or `python -c 'import __graft_entry__ as g; g.build()'` from the repository root).")
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().mdf_last_error().decode("utf-8", "replace")


# include/mdfri.h MDF_DEFAULT_CHUNK_ROWS, stated here too so that host-only callers (sharding.plan_summary, bench.py --dry-plan, the
# binding's keyword default) need not load the library -- and with it the HIP runtime -- to read a constant; tests/test_abi_cpu.py
# asserts that header, library, this constant, bench.py and the compiled binding all say the same number.
DEFAULT_CHUNK_ROWS = 262144


def default_chunk_rows() -> int:
    """Residue rows per fused chunk where the caller does not say (include/mdfri.h MDF_DEFAULT_CHUNK_ROWS), as the LOADED library states it."""
    return int(lib().mdf_default_chunk_rows())


def check(rc: int) -> None:
    if rc == MDF_OK:
        return
    msg = last_error()
    if rc == MDF_EINVAL or rc == MDF_EBADCHAR:
        raise ValueError(msg)
    if rc == MDF_ENOMEM:
        raise MemoryError(msg)
    if rc == MDF_ECAPACITY:
        raise CapacityError(rc, msg)
    if rc == MDF_EIO:
        raise OSError(msg)
    raise MdfriError(rc, msg)


def ptr(a):
    """Raw address of a NumPy array or a torch tensor (None -> NULL)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return c_void_p(a.data_ptr())
    return c_void_p(a.ctypes.data)


def device_count() -> int:
    return int(lib().mdf_device_count())


def current_device() -> int:
    return int(lib().mdf_current_device())
