"""ctypes binding of include/cirkit_hip.h.  Fails loudly: there is no CPU or eager fallback --
if the HIP extension is missing or a call returns an error status, an exception is raised."""

from __future__ import annotations

import ctypes as C
import os
from typing import Any

# CIRKIT_HIP_LIB: a lab build of the same library (scripts/lab_build.sh, scripts/defect_injection.sh); never a different backend
_LIB_PATH = os.environ.get("CIRKIT_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcirkit_hip.so")

ABI_VERSION = 48

CK_SUM_CAT = 0
CK_SUM_PROD = 1
CK_SUM_KRON = 2
CK_W_ROWMAJOR = 0
CK_W_TILED_F32 = 1
CK_UNARY_SIGMOID = 0
CK_UNARY_SCALED_SIGMOID = 1
CK_UNARY_EXP = 2
CK_UNARY_LOG = 3
CK_UNARY_SQUARE = 4
CK_UNARY_CLAMP = 5
CK_UNARY_SOFTPLUS = 6

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float

class SoftmaxJob(C.Structure):
    _fields_ = [
        ("inp", C.c_void_p),
        ("out", C.c_void_p),
        ("rows", C.c_int64),
        ("len", C.c_int32),
        ("k", C.c_int32),
        ("kind", C.c_int32),
        ("block_begin", C.c_int32),
        ("in2", C.c_void_p),
        ("idx", C.c_void_p),
        ("out2", C.c_void_p),
    ]


class EinsumDesc(C.Structure):
    """ck_einsum_desc of include/cirkit_hip.h."""

    _fields_ = [
        ("x", C.c_void_p * 4),
        ("out", C.c_void_p),
        ("n_ops", C.c_int32), ("n_idx", C.c_int32), ("n_out", C.c_int32), ("F", C.c_int32), ("out_complex", C.c_int32),
        ("is_complex", C.c_int32 * 4),
        ("extent", C.c_int32 * 8),
        ("stride", (C.c_int64 * 8) * 4),
        ("fold_stride", C.c_int64 * 4),
    ]


class LeafLaunch(C.Structure):
    """ck_leaf_launch of include/cirkit_hip.h."""

    _fields_ = [
        ("table", C.c_void_p),
        ("table_scale", C.c_void_p),
        ("xt", C.c_void_p),
        ("scope", C.c_void_p),
        ("w_levels", C.POINTER(C.c_void_p)),
        ("nodes", C.c_void_p),
        ("node_off", C.POINTER(C.c_int32)),
        ("leaf_off", C.c_int32),
        ("n_seg", C.c_int32),
        ("out", C.c_void_p),
        ("work", C.c_void_p),
        ("n_wg", C.c_int32), ("waves", C.c_int32), ("depth", C.c_int32), ("B", C.c_int32), ("K", C.c_int32),
        ("C", C.c_int32), ("preclamped", C.c_int32), ("w_layout", C.c_int32),
        ("signed_redo", C.c_void_p),
        ("n_roots", C.c_int32),
        ("x_input", C.c_int32),
        ("x_rows", C.c_void_p),
        ("bad_input", C.c_void_p),
        ("D", C.c_int32),
        ("contraction", C.c_int32), ("reserved2", C.c_int32),
        ("keep_levels", C.POINTER(C.c_void_p)),
        ("keep_redo", C.c_void_p),
        ("x_pairs", C.c_int32),
        ("root_tab", C.c_void_p),
        ("signs_out", C.c_void_p),
    ]


class LeafBwdLaunch(C.Structure):
    """ck_leaf_bwd_launch of include/cirkit_hip.h."""

    _fields_ = [
        ("unit_tab", C.c_void_p),
        ("work", C.c_void_p),
        ("n_seg", C.c_int32), ("n_wg", C.c_int32), ("B", C.c_int32), ("C", C.c_int32), ("D", C.c_int32), ("leaf", C.c_int32),
        ("waves", C.c_int32), ("gin_rowmajor", C.c_int32),
        ("gin", C.c_void_p),
        ("y_p", C.c_void_p),
        ("y_c", C.c_void_p),
        ("table", C.c_void_p),
        ("x_rows", C.c_void_p),
        ("w_p", C.c_void_p),
        ("w_q", C.c_void_p),
        ("dw_p", C.c_void_p),
        ("dw_q", C.c_void_p),
        ("gout", C.c_void_p),
        ("redo", C.c_void_p),
        ("is_signed", C.c_int32), ("reserved", C.c_int32),
    ]


class TailParamsLaunch(C.Structure):
    """ck_tail_params_launch of include/cirkit_hip.h."""

    _fields_ = [
        ("folds", C.c_void_p),
        ("level_begin", C.c_void_p),
        ("n_folds", C.c_int32), ("n_levels", C.c_int32), ("n_slots", C.c_int32), ("B", C.c_int32), ("w_layout", C.c_int32),
        ("C", C.c_int32),
        ("ll", C.c_void_p),
        ("ll_partial", C.c_void_p),
        ("ll_ticket", C.c_void_p),
        ("bad_input", C.c_void_p),
        ("cat_logits", C.c_void_p),
        ("cat_idx", C.c_void_p),
        ("dense_logits", C.c_void_p),
        ("table", C.c_void_p),
        ("table_scale", C.c_void_p),
        ("rows", C.c_void_p),
        ("n_tables", C.c_int32), ("n_rows", C.c_int32),
        ("ll_cell", C.c_int32),
    ]


class TableOpt(C.Structure):
    """ck_table_opt of include/cirkit_hip.h."""

    _fields_ = [("state", C.c_void_p), ("m1_cat", C.c_void_p), ("m2_cat", C.c_void_p), ("m1_dense", C.c_void_p), ("m2_dense", C.c_void_p),
                ("table", C.c_void_p), ("table_scale", C.c_void_p)]


class OptState(C.Structure):
    """ck_opt_state of include/cirkit_hip.h (a DEVICE struct: this mirror is for building its initial bytes)."""

    _fields_ = [("lr", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float), ("bc1", C.c_float), ("bc2", C.c_float),
                ("step", C.c_int32), ("skipped", C.c_int32), ("skip_now", C.c_int32), ("kind", C.c_int32),
                ("b1d", C.c_double), ("b2d", C.c_double)]


class RootLaunch(C.Structure):
    """ck_root_launch of include/cirkit_hip.h."""

    _fields_ = [
        ("pool", C.c_void_p), ("in_off", C.c_void_p), ("n_in", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p), ("out", C.c_void_p),
        ("gx", C.c_void_p), ("seed", C.c_void_p), ("ll", C.c_void_p), ("part", C.c_void_p), ("ticket", C.c_void_p),
        ("dtheta_w", C.c_void_p), ("dtheta_c", C.c_void_p), ("theta_w", C.c_void_p), ("m1_w", C.c_void_p), ("m2_w", C.c_void_p),
        ("w_out", C.c_void_p), ("theta_c", C.c_void_p), ("m1_c", C.c_void_p), ("m2_c", C.c_void_p), ("c_out", C.c_void_p),
        ("opt", C.c_void_p), ("bad_flag", C.c_void_p),
        ("seed_const", C.c_float), ("R", C.c_int32), ("B", C.c_int32), ("mode", C.c_int32), ("n_wg", C.c_int32), ("S", C.c_int32),
    ]


# numpy mirrors of the DEVICE job tables (ck_sum_job / ck_mix_job: 128 bytes; ck_nsum_job: 16 bytes)
SUM_JOB_DTYPE = [("w", "<u8"), ("out", "<u8"), ("gx", "<u8"), ("dtheta", "<u8"), ("theta", "<u8"), ("m1", "<u8"), ("m2", "<u8"),
                 ("w_out", "<u8"), ("part", "<u8"), ("ticket", "<u8"), ("in_off", "<i4"), ("n_in", "<i4"), ("g_off", "<i4"),
                 ("n_g", "<i4"), ("row0", "<i4"), ("row1", "<i4"), ("split", "<i4"), ("n_split", "<i4"), ("mode", "<i4"),
                 ("C", "<i4"), ("xrow", "<u8"), ("mix_out", "<u8"), ("mix_w", "<u8"), ("mix_dw", "<u8"), ("partner_off", "<i4"),
                 ("n_partner", "<i4"), ("mix_h", "<i4"), ("mix_H", "<i4"), ("reserved_a", "<i8"), ("reserved_b", "<i8"), ("reserved_c", "<i8")]
MIX_JOB_DTYPE = [("w", "<u8"), ("out", "<u8"), ("gx", "<u8"), ("dtheta", "<u8"), ("theta", "<u8"), ("m1", "<u8"), ("m2", "<u8"),
                 ("w_out", "<u8"), ("part", "<u8"), ("ticket", "<u8"), ("in_off", "<i4"), ("H", "<i4"), ("g_off", "<i4"),
                 ("n_g", "<i4"), ("row0", "<i4"), ("row1", "<i4"), ("split", "<i4"), ("n_split", "<i4"), ("mode", "<i4"),
                 ("S", "<i4"), ("reserved1", "<i8")]
NSUM_JOB_DTYPE = [("out", "<u8"), ("in_off", "<i4"), ("n_in", "<i4")]
CAT_JOB_DTYPE = [("x", "<u8"), ("theta", "<u8"), ("table", "<u8"), ("dtheta", "<u8"), ("theta_out", "<u8"), ("m1", "<u8"), ("m2", "<u8"),
                 ("table_out", "<u8"), ("g_off", "<i4"), ("n_g", "<i4"), ("mode", "<i4"), ("reserved", "<i4")]
GAUSS_JOB_DTYPE = [("mean", "<u8"), ("stddev", "<u8"), ("x", "<u8"), ("dmean", "<u8"), ("dsd", "<u8"), ("th_mean", "<u8"), ("m1_mean", "<u8"),
                   ("m2_mean", "<u8"), ("th_sd", "<u8"), ("m1_sd", "<u8"), ("m2_sd", "<u8"), ("mean_out", "<u8"), ("sd_out", "<u8"),
                   ("g_off", "<i4"), ("n_g", "<i4"), ("vmin", "<f4"), ("vmax", "<f4"), ("has_ss", "<i4"), ("mode", "<i4")]


# name -> argtypes (restype is always int unless listed in _RESTYPES)
SIGNATURES: dict[str, list[Any]] = {
    "ck_abi_version": [],
    "ck_param_reduce": [_i, _p, _p, _l, _i, _l, _p],
    "ck_param_reduce_bwd": [_i, _p, _p, _p, _p, _l, _i, _l, _p],
    "ck_param_outer_sum": [_p, _p, _p, _l, _i, _i, _l, _p],
    "ck_param_outer_sum_bwd": [_p, _p, _l, _i, _i, _l, _i, _p],
    "ck_clin_table": [_p, _i, _p, _p, _i, _i, _p],
    "ck_clin_tail_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "ck_clin_leaf_fwd": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "ck_clin_layer_fwd": [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_comm_load": [C.c_char_p],
    "ck_comm_unique_id": [_p],
    "ck_comm_init": [_p, _i, _i, _i, C.POINTER(C.c_void_p)],
    "ck_comm_all_reduce_f64": [_p, _p, _l, _p],
    "ck_comm_all_reduce_f32": [_p, _p, _l, _p],
    "ck_comm_all_reduce_async_f64": [_p, _p, _l, _p],
    "ck_comm_wait": [_p, _p],
    "ck_comm_info": [_p, C.POINTER(C.c_int32), C.c_char_p, _i],
    "ck_comm_destroy": [_p],
    "ck_device_info": [_i, C.POINTER(_l)],
    "ck_transpose_i64_to_i32": [_p, _p, _i, _i, _p],
    "ck_transpose_f32": [_p, _p, _i, _i, _p],
    "ck_stage_categories": [_p, _p, _i, _i, _p, _p, _i, _p, _p],
    "ck_poison_outputs": [_p, _l, _p, _p],
    "ck_zero_if_flag": [_p, _l, _p, _p],
    "ck_categorical_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_gaussian_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_gaussian_prod_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_embedding_clog_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_embedding_clog_c_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_embedding_log_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_categorical_clog_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_lse_to_clse": [_p, _p, _l, _p],
    "ck_constant_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_sum_lse_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_sum_lse_fwd_v": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_tucker_logits_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_tucker_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_debug_force_generic": [_i],
    "ck_sum_lse_fwd_c": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_sum_clse_gather_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_mixing_lse_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_cp_lse_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_region_lse_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    "ck_cp_lse_fwd_v": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_region_lse_fwd_v": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_hadamard_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_kronecker_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_tensordot_lse_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_tensordot_lse_fwd_c": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_subtree_cat_cpt_fwd": [_p, _p, _p, _p, _p, C.POINTER(_p), _p, C.POINTER(C.c_int32), _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_leaf_persistent_fwd": [_p, _p, _p, _p, C.POINTER(_p), _p, C.POINTER(C.c_int32), _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "ck_leaf_walk_fwd": [C.POINTER(LeafLaunch), _p],
    "ck_tail_params_fwd": [C.POINTER(TailParamsLaunch), _p],
    "ck_tail16_lse_fwd": [_p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p],
    "ck_param_softmax": [_p, _p, _l, _i, _l, _i, _p],
    "ck_param_softmax_batch": [C.POINTER(SoftmaxJob), _i, _p],
    "ck_param_unary": [_i, _p, _p, _l, _f, _f, _p],
    "ck_param_gather_folds": [_p, _p, _p, _l, _l, _p],
    "ck_param_conj": [_p, _p, _l, _p],
    "ck_param_mixing_weight": [_p, _p, _i, _i, _i, _p],
    "ck_param_bmm": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_param_einsum": [_p, _p],
    "ck_param_transpose_last2": [_p, _p, _l, _i, _i, _i, _i, _p],
    "ck_param_transpose_last2_c": [_p, _p, _l, _i, _i, _i, _p],
    "ck_param_table_integral_row": [_p, _i, _i, _i, _i, _p],
    "ck_fill_f32": [_p, _l, _f, _p],
    "ck_sum_lse_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_sum_lse_bwd_c": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_debug_force_generic_bwd": [_i],
    "ck_hadamard_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_kronecker_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_gaussian_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "ck_mixing_lse_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ck_param_scaled_sigmoid_bwd": [_p, _p, _p, _l, _f, _f, _i, _p],
    "ck_param_softmax_bwd_strided": [_p, _p, _p, _l, _i, _l, _i, _i, _p],
    "ck_param_unary_bwd": [_i, _p, _p, _p, _p, _l, _i, _p],
    "ck_param_mixing_weight_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "ck_axpy_f32": [_p, _p, _f, _l, _p],
    "ck_param_binomial_table": [_p, _i, _p, _l, _i, _i, _p],
    "ck_param_gaussian_product_ms": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "ck_param_gaussian_product_ms_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "ck_param_gaussian_product_logz": [_p, _p, _p, _p, _p, _l, _i, _i, _p],
    "ck_param_gaussian_product_logz_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _p],
    "ck_segment_add_rows": [_p, _p, _p, _p, _p, _i, _l, _p],
    "ck_param_scatter_add_folds": [_p, _p, _p, _l, _l, _p],
    "ck_categorical_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "ck_tail_bwd": [_p, _i, _p, _i, _i, C.c_int64, _p],
    "ck_param_softmax_bwd_batch": [_p, _i, _i, _p, _p],
    "ck_fill_latch": [_p, _l, _f, _p, _p, _p, _p, _p],
    "ck_leaf_walk_bwd": [C.POINTER(LeafBwdLaunch), _p],
    "ck_table_dense_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, C.POINTER(TableOpt), _p],
    "ck_leaf_walk_bwd_redo": [_p, _p, _p, _i, _i, _i, _p, C.POINTER(C.c_int32), _i, _p, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _p, _p, _p, _i, _p, _i, _p],
    "ck_param_softmax_bwd": [_p, _p, _p, _l, _i, _i, _p],
    "ck_param_log_table_bwd": [_p, _p, _p, _i, _i, _i, _i, _p],
    "ck_adam_step": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _i, _f, _p, _p, _p],
    "ck_sgd_step": [_p, _p, _l, _f, _f, _p, _p],
    "ck_copy_strided_f32": [_p, _p, _l, _l, _l, _p],
    "ck_fill_strided_f32": [_p, _l, _l, _f, _p],
    "ck_embedding_weight_bwd": [_p, _p, _p, _i, _i, _i, _p],
    "ck_squared_ll": [_p, _l, _l, _p, _p, _p],
    "ck_embedding_bwd": [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ck_tensordot_lse_fwd_h": [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_tensordot2_lse_fwd": [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_tensordot_lse_bwd": [_p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ck_tensordot2_lse_bwd": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ck_slse_table": [_p, _p, _p, _l, _p],
    "ck_slse_tables": [_p, _p, _p, _p, _i, _i, _p],
    "ck_slse_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "ck_slse_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "ck_latch_flag": [_p, _p, _p],
    "ck_jobs_sum64_fwd": [_p, _i, _p, _p],
    "ck_jobs_sum64_bwd": [_p, _i, _p, _p, _i, _p],
    "ck_jobs_mix_params": [_p, _i, _p, _p],
    "ck_jobs_mix_fwd": [_p, _i, _p, _i, _p],
    "ck_jobs_mix_bwd": [_p, _i, _p, _i, _l, _p, _p],
    "ck_jobs_nsum": [_p, _i, _p, _l, _p],
    "ck_jobs_root": [C.POINTER(RootLaunch), _p],
    "ck_jobs_cat_bwd": [_p, _i, _p, _i, _i, _p, _p],
    "ck_jobs_gauss_bwd": [_p, _i, _p, _i, _p, _p],
    "ck_opt_step_range": [_p, _p, _p, _p, _p, _l, _p, _p],
    "ck_opt_tick": [_p, _p, _p, _p],
    "ck_ll_sum": [_p, _l, _l, _p, _p],
    "ck_program_begin": [C.POINTER(_p)],
    "ck_program_end": [_p],
    "ck_program_num_ops": [_p],
    "ck_program_launch": [_p, _i, _p],
    "ck_program_set_input": [_p, _i, _p],
    "ck_set_workspace": [_p, _l],
    "ck_program_destroy": [_p],
}
_RESTYPES = {"ck_last_error": C.c_char_p}


class HipExtensionError(RuntimeError):
    """The HIP extension is missing, stale, or returned an error status."""


_lib: C.CDLL | None = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libcirkit_hip.so (built by `python -m cirkit_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch provides the device storage and must bring up ITS HIP runtime first: the library then
    # binds to the libamdhip64 already in the process instead of loading a second copy.
    import torch  # noqa: F401

    if not os.path.exists(_LIB_PATH):
        raise HipExtensionError(
            f"{_LIB_PATH} not found: build it with `python -m cirkit_amd.build` "
            "(there is no CPU fallback for the evaluation path)"
        )
    lib = C.CDLL(_LIB_PATH)
    lib.ck_last_error.restype = C.c_char_p
    lib.ck_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    got = lib.ck_abi_version()
    if got != ABI_VERSION:
        raise HipExtensionError(f"libcirkit_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().ck_last_error().decode("utf-8", "replace")
        if status == -1:
            raise ValueError(f"{what}: {msg}")
        if status == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise HipExtensionError(f"{what}: status {status}: {msg}")


def call(name: str, *args: Any) -> None:
    """Call an entry point and raise on a non-zero status.  ValueError on CK_ERR_INVALID mirrors the
    reference's shape errors (e.g. TorchSumLayer.__init__, layers/inner.py:237-242)."""
    check(getattr(load(), name)(*args), name)
