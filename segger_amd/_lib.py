"""ctypes binding of libsegger_amd.so (C ABI in include/segger_amd.h).

There is NO fallback: if the shared library is missing or a call fails, the
product raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C segger_amd/csrc -j8``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must be imported first: the library binds to torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEGGER_AMD_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("SEGGER_AMD_LIB") or os.path.join(_HERE, "libsegger_amd.so")
ABI_VERSION = 30

SEGGER_F32, SEGGER_BF16, SEGGER_F16 = 0, 1, 2
DTYPE_CODE = {torch.float32: SEGGER_F32, torch.bfloat16: SEGGER_BF16, torch.float16: SEGGER_F16}


class SeggerAmdError(RuntimeError):
    pass


c_i64p = C.c_void_p
vp = C.c_void_p


class Csr(C.Structure):
    _fields_ = [("indptr", vp), ("col", vp), ("eid", vp),
                ("n_rows", C.c_int64), ("n_cols", C.c_int64), ("n_edges", C.c_int64), ("row_order", vp),
                ("blk_cnt", vp), ("blk_src", vp), ("col_local", vp)]


class GatFwdArgs(C.Structure):
    _fields_ = [
        ("by_dst", Csr),
        ("x_l", vp), ("ld_xl", C.c_int64),
        ("x_r", vp), ("ld_xr", C.c_int64),
        ("att", vp), ("bias", vp),
        ("heads", C.c_int32), ("channels", C.c_int32), ("dtype", C.c_int32), ("apply_gelu", C.c_int32),
        ("negative_slope", C.c_float), ("dropout_p", C.c_float), ("seed", C.c_uint64), ("seed_dev", vp),
        ("out", vp), ("ld_out", C.c_int64),
        ("pre", vp), ("ld_pre", C.c_int64),
        ("lse", vp), ("alpha", vp),
        ("keep_bits", vp),
    ]


class GatBwdArgs(C.Structure):
    _fields_ = [
        ("by_dst", Csr), ("by_src", Csr),
        ("x_l", vp), ("ld_xl", C.c_int64),
        ("x_r", vp), ("ld_xr", C.c_int64),
        ("att", vp), ("bias", vp),
        ("heads", C.c_int32), ("channels", C.c_int32), ("dtype", C.c_int32), ("apply_gelu", C.c_int32),
        ("negative_slope", C.c_float), ("dropout_p", C.c_float), ("seed", C.c_uint64), ("seed_dev", vp),
        ("grad_out", vp), ("ld_go", C.c_int64),
        ("pre", vp), ("ld_pre", C.c_int64),
        ("lse", vp),
        ("grad_pre", vp), ("ld_gp", C.c_int64),
        ("dsum", vp),
        ("grad_xl", vp), ("ld_gxl", C.c_int64),
        ("grad_xr", vp), ("ld_gxr", C.c_int64),
        ("grad_att", vp), ("grad_bias", vp),
        ("workspace", vp), ("workspace_bytes", C.c_size_t),
        ("keep_bits_dst", vp), ("keep_bits_src", vp),
        ("src_unique", C.c_int32),
        ("zero_rows_out", vp), ("ld_zero", C.c_int64), ("grad_xl_zeroed", C.c_int32), ("passes", C.c_int32),
    ]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("step", C.c_void_p), ("numel", C.c_int64)]


class StageSeg(C.Structure):
    _fields_ = [
        ("dst", vp), ("src", vp),
        ("n_copy", C.c_int64), ("n_total", C.c_int64),
        ("a", C.c_int64), ("b", C.c_int64), ("c", C.c_int64),
        ("dst_bytes", C.c_int32), ("src_bytes", C.c_int32),
        ("fill", C.c_int32), ("copy_add", C.c_int32),
    ]


class PlanesJob(C.Structure):
    _fields_ = [("w", vp), ("rows", C.c_int32), ("cols", C.c_int32), ("transpose", C.c_int32), ("reserved_", C.c_int32),
                ("planes", vp)]


PLANES_MAX_JOBS = 32        # SEGGER_PLANES_MAX_JOBS


class TransposeSeg(C.Structure):
    _fields_ = [("dst", vp), ("src", vp), ("rows", C.c_int32), ("cols", C.c_int32)]


class PackSeg(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("dst_t", vp), ("ld_t", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("dst_f32", C.c_int32), ("reserved_", C.c_int32)]


FILL_CONST, FILL_TILE, FILL_DIV, FILL_MOD, FILL_RAMP = range(5)


class EdgeArgmaxArgs(C.Structure):
    _fields_ = [
        ("by_src", Csr),
        ("z_src", vp), ("ld_zs", C.c_int64),
        ("z_dst", vp), ("ld_zd", C.c_int64),
        ("channels", C.c_int32), ("dtype", C.c_int32),
        ("eps", C.c_float), ("use_min_similarity", C.c_int32), ("min_similarity", C.c_float),
        ("dst_index", vp),
        ("max_sim", vp), ("max_eid", vp), ("seg_idx", vp), ("sim", vp),
    ]


class BitsJob(C.Structure):
    _fields_ = [("eid", vp), ("n_edges", C.c_int64), ("seeds", vp), ("n_seeds", C.c_int32), ("reserved_", C.c_int32),
                ("bits", vp), ("plane_stride", C.c_int64)]


class SampleJob(C.Structure):
    _fields_ = [("lab", vp), ("n", C.c_int64), ("n_clusters", C.c_int32), ("reserved_", C.c_int32),
                ("cdf_pos", vp), ("cdf_neg", vp), ("counts", vp), ("offsets", vp), ("members", vp),
                ("seed", C.c_uint64), ("dists", vp), ("pos", vp), ("neg", vp), ("d_pos", vp), ("d_neg", vp)]


class StepDrawsArgs(C.Structure):
    _fields_ = [("bits", vp), ("n_bits", C.c_int32), ("heads", C.c_int32), ("dropout_p", C.c_float), ("n_samplers", C.c_int32),
                ("samplers", SampleJob * 2),
                ("neg_pos", vp), ("neg_n", C.c_int64), ("neg_n_b", C.c_int64), ("neg_n_b_dev", vp), ("neg_seed", C.c_uint64),
                ("neg_out", vp), ("seed_dev", vp), ("advance", vp), ("n_advance", C.c_int32), ("reserved_", C.c_int32)]


class TripletArgs(C.Structure):
    _fields_ = [
        ("src", vp), ("pos", vp), ("neg", vp), ("n_edges", C.c_int64),
        ("z_a", vp), ("ld_za", C.c_int64), ("n_a", C.c_int64),
        ("z_b", vp), ("ld_zb", C.c_int64), ("n_b", C.c_int64),
        ("channels", C.c_int32), ("dtype", C.c_int32),
        ("margin", C.c_float), ("eps", C.c_float),
        ("loss", vp), ("grad_scale", C.c_float), ("grad_scale_dev", vp),
        ("grad_a", vp), ("grad_b", vp), ("grad_a_packed", C.c_int32), ("grad_b_packed", C.c_int32),
        ("contrib", vp),
        ("workspace", vp), ("workspace_bytes", C.c_size_t),
        ("pos_indptr", vp), ("pos_eid", vp), ("anchor_unique", C.c_int32), ("loss_kind", C.c_int32),
        ("grad_a_rows", vp),
    ]


class L2NormSeg(C.Structure):
    _fields_ = [("y", vp), ("ld_y", C.c_int64), ("n", C.c_int64), ("out", vp), ("ld_out", C.c_int64),
                ("gz", vp), ("ld_gz", C.c_int64), ("gz_f32", C.c_int32), ("reserved_", C.c_int32)]


class GeneTableArgs(C.Structure):
    _fields_ = [
        ("table", vp), ("n_genes", C.c_int32), ("D", C.c_int32), ("n_w", C.c_int32), ("dtype", C.c_int32),
        ("w", vp * 4), ("ld_w", C.c_int64 * 4), ("m", C.c_int32 * 4), ("b", vp * 4),
        ("tab", vp), ("ld_tab", C.c_int64), ("wc", vp), ("wc_t", vp),
        ("g_tab", vp), ("g_wc", vp),
        ("g_table", vp), ("g_w", vp * 4), ("g_b", vp * 4),
    ]


class FrontJoinArgs(C.Structure):
    _fields_ = [
        ("table", vp), ("ids", vp), ("n_rows_table", C.c_int32), ("D", C.c_int32), ("dtype", C.c_int32), ("reserved_", C.c_int32),
        ("pe", vp), ("ld_pe", C.c_int64), ("n_tx", C.c_int64), ("n_bd", C.c_int64),
        ("xb", vp), ("ld_xb", C.c_int64),
        ("out_tx", vp), ("ld_out_tx", C.c_int64), ("out_bd", vp), ("ld_out_bd", C.c_int64),
        ("g_tx", vp), ("ld_g_tx", C.c_int64), ("g_bd", vp), ("ld_g_bd", C.c_int64),
        ("g_pe", vp), ("ld_g_pe", C.c_int64), ("g_xb", vp), ("ld_g_xb", C.c_int64),
        ("g_table", vp), ("gene_ptr", vp), ("gene_rows", vp), ("workspace", vp), ("workspace_bytes", C.c_size_t),
    ]


class LossHeadArgs(C.Structure):
    _fields_ = [
        ("z_tx", vp), ("ld_ztx", C.c_int64), ("n_tx", C.c_int64),
        ("z_bd", vp), ("ld_zbd", C.c_int64), ("n_bd", C.c_int64),
        ("channels", C.c_int32), ("dtype", C.c_int32),
        ("tx_pos", vp), ("tx_neg", vp), ("tx_margin", C.c_float), ("tx_eps", C.c_float),
        ("bd_pos", vp), ("bd_neg", vp), ("bd_dpos", vp), ("bd_dneg", vp), ("bd_w", vp), ("bd_eps", C.c_float),
        ("sg_kind", C.c_int32),
        ("sg_src", vp), ("sg_pos", vp), ("sg_neg", vp), ("n_sg", C.c_int64), ("sg_margin", C.c_float), ("sg_eps", C.c_float),
        ("sg_pos_indptr", vp), ("sg_pos_eid", vp), ("sg_of_tx", vp),
        ("a", vp), ("b", vp), ("grad_out", vp), ("out", vp), ("grad_raw", vp),
        ("tx_w", vp), ("tx_state", vp), ("tx_next", vp), ("tx_hot_id", vp), ("tx_hot_acc", vp),
        ("y_tx", vp), ("ld_ytx", C.c_int64), ("norm_eps", C.c_float), ("reserved_", C.c_int32),
        ("grad_tx", vp), ("ld_gtx", C.c_int64), ("grad_bd", vp),
        ("workspace", vp), ("workspace_bytes", C.c_size_t), ("ticket", vp),
    ]


class LinearArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", C.c_int64), ("w", vp), ("bias", vp), ("y", vp), ("ldy", C.c_int64),
                ("n_rows", C.c_int64), ("m_out", C.c_int32), ("reserved_", C.c_int32)]


class WgradArgs(C.Structure):
    _fields_ = [("dy", vp), ("ld_dy", C.c_int64), ("x", vp), ("ld_x", C.c_int64), ("w_t", vp), ("n_rows", C.c_int64),
                ("m_out", C.c_int32), ("reserved_", C.c_int32), ("grad_w", vp), ("grad_b", vp), ("dx", vp),
                ("ld_dx", C.c_int64), ("workspace", vp), ("workspace_bytes", C.c_size_t)]


# every symbol include/segger_amd.h declares: name -> (restype, argtypes)
EXPORTS = {
    "segger_abi_version": (C.c_int, []),
    "segger_csr_block_tables": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp]),
    "segger_last_error": (C.c_char_p, []),
    "segger_csr_from_coo_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "segger_csr_from_coo": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "segger_csr_row_order": (C.c_int, [vp, C.c_int64, C.c_int32, vp, vp]),
    "segger_gatv2_fwd": (C.c_int, [C.POINTER(GatFwdArgs), vp]),
    "segger_gatv2_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "segger_gatv2_bwd": (C.c_int, [C.POINTER(GatBwdArgs), vp]),
    "segger_dropout_bits": (C.c_int, [vp, C.c_int64, C.c_int32, C.c_float, vp, C.c_int32, vp, vp, C.c_int64, vp]),
    "segger_gatv2_has_specialised": (C.c_int, [C.c_int32, C.c_int32]),
    "segger_coo_unique": (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp, vp]),
    "segger_stage": (C.c_int, [C.POINTER(StageSeg), C.c_int32, vp]),
    "segger_adam_step": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, vp]),
    "segger_adam_step_ex": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, vp, C.c_int64, vp]),
    "segger_adam_step_dev": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, vp, C.c_int32, vp, C.c_int64, vp]),
    "segger_transpose_many": (C.c_int, [C.POINTER(TransposeSeg), C.c_int32, vp]),
    "segger_posmlp_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "segger_posmlp_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                    C.c_int32, C.c_int32, vp]),
    "segger_posmlp_wgrad": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_float, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "segger_sample_negatives": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_uint64, vp, vp, vp]),
    "segger_edge_cos_argmax": (C.c_int, [C.POINTER(EdgeArgmaxArgs), vp]),
    "segger_triplet_sample": (C.c_int, [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]),
    "segger_metric_fwd": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_float, vp, vp,
                                    C.c_size_t, vp]),
    "segger_metric_bwd": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp]),
    "segger_triplet_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "segger_triplet_fwd": (C.c_int, [C.POINTER(TripletArgs), vp]),
    "segger_triplet_bwd": (C.c_int, [C.POINTER(TripletArgs), vp]),
    "segger_segment_minmax": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, vp, vp]),
    "segger_linear_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "segger_linear_fwd": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_f32_split_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "segger_linear_fwd_f32_split": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_pair": (C.c_int, [C.POINTER(LinearArgs), C.POINTER(LinearArgs), C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_pair_k": (C.c_int, [C.POINTER(LinearArgs), C.c_int32, C.POINTER(LinearArgs), C.c_int32, C.c_int32, vp]),
    "segger_linear_wgrad_pair": (C.c_int, [C.POINTER(WgradArgs), C.POINTER(WgradArgs), C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_silu_grad": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                              C.c_int32, vp]),
    "segger_linear_fwd_rowbias": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                            C.c_int32, vp]),
    "segger_linear_wgrad_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "segger_linear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "segger_linear_wgrad": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, vp,
                                      vp, C.c_size_t, vp]),
    "segger_loss_combine_fwd": (C.c_int, [vp, vp, vp, C.c_int32, vp, vp]),
    "segger_loss_combine_bwd": (C.c_int, [vp, vp, vp, C.c_int32, vp, vp]),
    "segger_reductions_defer_begin": (C.c_int, []),
    "segger_reductions_pending": (C.c_int, []),
    "segger_reductions_flush": (C.c_int, [vp]),
    "segger_linear_wgrad_dx_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "segger_l2norm_bwd2": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_float, vp, C.c_int64,
                                     C.c_int32, vp]),
    "segger_gatv2_fwd_pair": (C.c_int, [vp, vp, vp]),
    "segger_gatv2_bwd_pair": (C.c_int, [vp, vp, vp]),
    "segger_pack_refresh": (C.c_int, [vp, C.c_int32, C.c_int32, vp]),
    "segger_dropout_bits_many": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_float, vp, vp]),
    "segger_step_advance": (C.c_int, [vp, C.c_int64, vp, vp]),
    "segger_step_draws": (C.c_int, [C.POINTER(StepDrawsArgs), vp]),
    "segger_segment_minmax_ex": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int32, vp]),
    "segger_triplet_partial_count": (C.c_int64, [C.c_int64]),
    "segger_loss_combine_partials_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, vp, vp]),
    "segger_posmlp_bwd_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "segger_posmlp_bwd": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_float, C.c_int32, vp, vp, vp, vp, vp,
                                    C.c_size_t, vp]),
    "segger_posmlp_bwd_pair_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "segger_posmlp_bwd_pair": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_float,
                                         C.c_int32, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "segger_linear_wgrad_dx_gate_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "segger_linear_wgrad_dx": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, vp,
                                         vp, C.c_int64, vp, C.c_int64, vp, C.c_size_t, vp]),
    "segger_posfreq": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_float, C.c_float, vp, C.c_int32, vp]),
    "segger_posenc_poly_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "segger_posenc_poly_coef": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_float, vp, vp]),
    "segger_posenc_poly_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_float, vp, C.c_int32, vp, vp, vp, vp]),
    "segger_posenc_poly_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "segger_posenc_poly_wgrad": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, vp, vp, vp, C.c_size_t, vp]),
    "segger_embed_gelu_fwd": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int64, C.c_int32, vp]),
    "segger_linear_fwd_f32_split_rowbias": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_f32_act": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, vp]),
    "segger_linear_fwd_f32_gate": (C.c_int, [vp, C.c_int64, vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp]),
    "segger_f32_split_planes": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]),
    "segger_f32_split_planes_many": (C.c_int, [vp, C.c_int32, vp]),
    "segger_linear_wgrad_f32_split_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "segger_linear_wgrad_f32_split": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "segger_gene_table_fwd": (C.c_int, [C.POINTER(GeneTableArgs), vp]),
    "segger_gene_table_bwd": (C.c_int, [C.POINTER(GeneTableArgs), vp]),
    "segger_front_join_fwd": (C.c_int, [C.POINTER(FrontJoinArgs), vp]),
    "segger_front_join_bwd": (C.c_int, [C.POINTER(FrontJoinArgs), vp]),
    "segger_embed_gelu_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "segger_embed_gelu_bwd": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int64,
                                        vp, vp, vp, vp, C.c_size_t, C.c_int32, vp]),
    "segger_l2norm_fwd": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_float, vp, C.c_int64, C.c_int32, vp]),
    "segger_l2norm_bwd": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32, C.c_float, vp, C.c_int64, C.c_int32, vp]),
    "segger_segment_rowsum_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "segger_segment_rowsum": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp,
                                        C.c_size_t, vp]),
    "segger_colsum_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "segger_colsum": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_size_t, vp]),
    "segger_l2norm_many": (C.c_int, [C.POINTER(L2NormSeg), C.c_int32, C.c_int32, C.c_float, C.c_int32, vp]),
    "segger_loss_head_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "segger_loss_head_max_hot_rows": (C.c_int64, [C.c_int64]),
    "segger_loss_head_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "segger_loss_head_fwd": (C.c_int, [C.POINTER(LossHeadArgs), vp]),
    "segger_loss_head_bwd": (C.c_int, [C.POINTER(LossHeadArgs), vp]),
    "segger_knn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "segger_knn_grid": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int32, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library once; raise loudly when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeggerAmdError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU/PyTorch "
            f"fallback for this path. Run `make -C {os.path.join(_HERE, 'csrc')} -j8`.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in EXPORTS.items():
        if not hasattr(lib, name):
            raise SeggerAmdError(f"{LIB_PATH} does not export {name}; rebuild it")
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.segger_abi_version()
    if v != ABI_VERSION:
        raise SeggerAmdError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().segger_last_error().decode("utf-8", "replace")
        raise SeggerAmdError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """``torch.cuda.device(device)`` only when it is not the current device already (the context manager costs
    ~10 us per launch; one process drives one GPU, so this is the common case)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device) -> int:
    """hipStream_t of torch's current stream on ``device`` (the raw-handle query: 0.3 us instead of 2.5 us for the Stream
    object -- an eager 1M-edge step makes ~110 launches and is host-bound)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SeggerAmdError(
                "segger_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                "(the CPU oracle under oracle/ is test infrastructure, not a code path)")
