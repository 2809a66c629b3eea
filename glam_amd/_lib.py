"""ctypes binding of ``libglam_hip.so`` (C ABI declared in ``include/glam_hip.h``).

The library is the product: if it is missing, or a tensor is not resident on a HIP device,
every op raises — there is no CPU or eager-PyTorch fallback behind these entry points.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLAM_HIP_LIB") or os.path.join(_HERE, "libglam_hip.so")   # override: kernel experiments (empty = unset)

GLAM_E_INVALID, GLAM_E_UNSUPPORTED, GLAM_E_HIP = -1, -2, -3

_vp, _i64, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

ABI_VERSION = 4     # = GLAM_ABI_VERSION of include/glam_hip.h: bumped with every change of an exported signature (kept next to SIGNATURES)

# name -> (restype, argtypes); mirrors include/glam_hip.h one to one
SIGNATURES = {
    "glam_abi_version": (_i32, []),
    "glam_last_error": (ctypes.c_char_p, []),
    "glam_route_enabled": (_i32, [ctypes.c_char_p]),
    "glam_prof_begin": (_i32, [_i32]),
    "glam_prof_end": (_i32, []),
    "glam_prof_read": (_i32, [_i32, ctypes.c_char_p, _i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)]),
    "glam_pad_group": (_i32, [_i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int32), _i32, _vp]),
    "glam_batch_fingerprint": (_i32, [_i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64), _vp, _vp]),
    "glam_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "glam_csr_build": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glam_batch_ptr": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "glam_triplet_fwd": (_i32, [_vp] * 8 + [_i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "glam_triplet_bwd_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "glam_triplet_bwd": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 6 + [_sz, _vp]),
    "glam_pool5_fwd": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "glam_pool5_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "glam_pool5_padded_fwd": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "glam_pool5_padded_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "glam_segment_pool_fwd": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "glam_segment_pool_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "glam_segment_attn_fwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "glam_segment_attn_bwd": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _vp, _vp, _vp]),
    "glam_edge_reduce_fwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "glam_edge_reduce_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "glam_edge_wsum_fwd": (_i32, [_vp] * 5 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "glam_edge_wsum_bwd": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "glam_edge_wsum_bwd_add": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "glam_pair_pool_workspace_bytes": (_sz, [_i64, _i32]),
    "glam_pair_pool_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glam_pair_pool_bwd": (_i32, [_vp] * 7 + [_i64, _i32, _vp, _vp, _vp]),
    "glam_pair_pool_add_supported": (_i32, [_i32]),
    "glam_pair_pool_bwd_add": (_i32, [_vp] * 7 + [_i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "glam_gru_tail_rng_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "glam_gru_tail_rng_bwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 6),
    "glam_bias_res_act_rng_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "glam_bias_res_act_rng_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "glam_ell_build": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "glam_triplet_fwd_ell_supported": (_i32, [_i32, _i32, _i32]),
    "glam_triplet_fwd_ell": (_i32, [_vp] * 7 + [_i64, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i32, _vp]),
    "glam_triplet_layer_ws_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "glam_triplet_layer_infer_supported": (_i32, [_i32, _i32, _i32]),
    "glam_triplet_layer_fwd_ell": (_i32, [_vp] * 5 + [_i32, _i64, _i64, _i32, _i32, _i32, _f32] + [_vp] * 5 + [_vp]),
    "glam_relation_mlp_supported": (_i32, [_i32, _i32, _i64]),
    "glam_relation_mlp_workspace_bytes": (_sz, [_i32, _i32, _i64]),
    "glam_relation_mlp_fwd": (_i32, [_vp] * 4 + [_i32, _i32, _i64, _vp, _vp, _vp]),
    "glam_relation_mlp_bwd": (_i32, [_vp] * 3 + [_i32, _i32, _i64] + [_vp] * 5 + [_sz, _vp]),
    "glam_pair_pool5_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "glam_gru_fused_supported": (_i32, [_i32]),
    "glam_gru_ws_supported": (_i32, [_i32]),
    "glam_gru_bwd_ws": (_i32, [_vp] * 9 + [_i64, _i32, _i32, _i32, _f32, _i32] + [_vp] * 5 + [_vp]),
    "glam_gru_bwd_ws_rng": (_i32, [_vp] * 10 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _i32] + [_vp] * 5 + [_vp]),
    "glam_gru_ws_fwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32] + [_vp] * 4 + [_vp]),
    "glam_gru_ws_rng_fwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 7 + [_vp]),
    "glam_gru_ws_fwd_xc": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32] + [_vp] * 5 + [_vp]),
    "glam_gru_ws_rng_fwd_xc": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 8 + [_vp]),
    "glam_gru_ws_pre_bytes": (_sz, []),
    "glam_gru_ws_make_pre": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "glam_gru_ws_fwd_pre": (_i32, [_vp] * 6 + [_i64, _i32, _i32, _i32, _f32] + [_vp] * 5 + [_vp]),
    "glam_gru_ws_rng_fwd_pre": (_i32, [_vp] * 6 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 8 + [_vp]),
    "glam_gru_ws_fwd_pre_node": (_i32, [_vp] * 6 + [_i64, _i32, _i32, _i32, _f32] + [_vp] * 5 + [_vp, _i32, _vp, _vp] + [_vp]),
    "glam_gru_ws_rng_fwd_pre_node": (_i32, [_vp] * 6 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 8 + [_vp, _i32, _vp, _vp] + [_vp]),
    "glam_gru_bwd_ws_pre": (_i32, [_vp] * 8 + [_i64, _i32, _i32, _i32, _f32, _i32] + [_vp] * 5 + [_vp]),
    "glam_gru_bwd_ws_rng_pre": (_i32, [_vp] * 9 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _i32] + [_vp] * 5 + [_vp]),
    "glam_gru_fused_image_bytes": (_sz, []),
    "glam_gru_fused_make_images": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "glam_gru_fused_fwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32] + [_vp] * 5),
    "glam_gru_fused_rng_fwd": (_i32, [_vp] * 7 + [_i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32] + [_vp] * 8),
    "glam_colsum_workspace_bytes": (_sz, [_i32]),
    "glam_colsum": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "glam_dense_gemm": (_i32, [_vp, _i64, _i64, _vp, _f32, _vp, _i64, _i64, _vp, _i32, _f32, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "glam_linear_dense_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp]),
    "glam_linear_dense_bwd": (_i32, [_vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "glam_dense_ws_bytes": (_sz, []),
    "glam_linear_dense_fwd_ws": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "glam_linear_dense_bwd_ws": (_i32, [_vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glam_wgrad_gemm_sets2": (_i32, [_i32, ctypes.POINTER(_vp), _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "glam_wgrad_gemm_sets": (_i32, [_i32, ctypes.POINTER(_vp), _i32, _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "glam_wgrad_gemm_pair_split_seg": (_i32, [_i32, ctypes.POINTER(_vp), _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i32, _vp, _vp, ctypes.POINTER(_vp), _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i32, _vp, _vp,
                                       _i64, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "glam_wgrad_gemm_gru_gates_seg": (_i32, [_i32, ctypes.POINTER(_vp), _i32, ctypes.POINTER(_vp), _i32, _i32, ctypes.POINTER(_vp), _i32, _vp, _vp, _vp, _vp,
                                      _i64, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "glam_wgrad_gemm_linear_sets": (_i32, [_i32, ctypes.POINTER(_vp), _i32, _i32, ctypes.POINTER(_vp), _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "glam_wgrad_gemm_linear": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "glam_wgrad_gemm_split": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "glam_wgrad_gemm_split_relu": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "glam_gru_make_images": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "glam_loss_workspace_bytes": (_sz, []),
    "glam_loss_fwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "glam_loss_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "glam_adam_max_tensors": (_i32, []),
    "glam_adam_step": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp] + [ctypes.c_double] * 5 + [_vp]),
    "glam_linear_narrow_supported": (_i32, [_i32, _i32]),
    "glam_linear_narrow_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "glam_linear_narrow_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "glam_linear_narrow_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glam_linear_narrow_act_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "glam_linear_narrow_act_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "glam_pair_pool5_bwd": (_i32, [_vp] * 7 + [_i64, _i32, _vp, _vp, _vp]),
    "glam_ts_gemm_relu_supported": (_i32, [_i32, _i32]),
    "glam_ts_gemm_relu": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    "glam_ts_gemm_rrelu_supported": (_i32, [_i32, _i32]),
    "glam_ts_gemm_act_node": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _i64, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "glam_ts_gemm_rrelu": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _i64, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "glam_ts_gemm_image_bytes": (_sz, [_i32, _i32]),
    "glam_ts_gemm_make_image": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "glam_ts_gemm_make_image_quad": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "glam_ts_gemm": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i64, _vp]),
    "glam_wgrad_workspace_bytes": (_sz, []),
    "glam_wgrad_gemm": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _vp, _i32, _i32, _vp, _sz, _vp]),
    "glam_wgrad_gemm_add": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "glam_graph_norm_fwd": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp]),
    "glam_graph_norm_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp]),
    "glam_graph_norm_bwd_add": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp, _vp]),
    "glam_graph_norm_drop_supported": (_i32, [_i64, _i64, _i32]),
    "glam_graph_norm_drop_fwd": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "glam_graph_norm_drop_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "glam_gru_gates_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "glam_wgrad_gemm_pair": (_i32, ([_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32] * 2) + [_i64, _vp, _sz, _vp]),
    "glam_ts_gemm_celu": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i64, _vp]),
    "glam_ts_gemm_add": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i64, _vp]),
    "glam_ts_gemm_pair": (_i32, ([_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32] * 2) + [_i64, _vp]),
    "glam_wgrad_gemm_pair_split": (_i32, ([_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp] * 2) + [_i64, _vp, _sz] + [_vp] * 5),
    "glam_gru_tail_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "glam_gru_tail_bwd": (_i32, [_vp] * 6 + [_i64, _i32, _i32, _f32] + [_vp] * 5),
    "glam_bias_res_act_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "glam_bias_res_act_bwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "glam_lstm_cell_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "glam_lstm_cell_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "glam_s2s_attn_fwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "glam_s2s_attn_bwd": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _vp, _vp, _vp]),
    "glam_gru_gates_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "glam_triplet_staged_floats": (_sz, [_i32, _i32, _i32]),
    "glam_triplet_staged_node_image": (_sz, [_i32, _i32, _i32]),
    "glam_triplet_staged_node_fragments": (_sz, [_i32, _i32, _i32]),
    "glam_triplet_dstaged_floats": (_sz, [_i32, _i32, _i32]),
    "glam_triplet_plain_floats": (_sz, [_i32, _i32, _i32]),
    "glam_triplet_stage_plain": (_i32, [_vp] * 5 + [_i32] * 5 + [_vp, _vp]),
    "glam_triplet_stage_params": (_i32, [_vp] * 5 + [_i32] * 5 + [_vp, _vp]),
    "glam_prestage": (_i32, [_vp] * 5 + [_i32] * 5 + [_vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_vp), _vp]),
    "glam_triplet_stage_params_bwd": (_i32, [_vp] * 4 + [_i32] * 5 + [_vp] * 5 + [_vp]),
    "glam_triplet_layer_fwd": (_i32, [_vp] * 6 + [_i64, _i64, _i32, _i32, _i32, _f32] + [_vp] * 5 + [_vp]),
    "glam_triplet_layer_bwd_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "glam_triplet_layer_bwd_params": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32] + [_vp] * 11 + [_sz, _vp]),
    "glam_triplet_layer_bwd_params_acc": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32] + [_vp] * 16 + [_i32, _vp, _vp, _sz, _vp]),
    "glam_triplet_layer_bwd_params_ell": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32] + [_vp] * 18 + [_i32, _vp, _vp, _sz, _vp]),
    "glam_triplet_layer_bwd_params_ell_add": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32] + [_vp] * 18 + [_i32, _vp, _vp, _sz, _vp, _vp]),
    "glam_triplet_layer_bwd_data_ell": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                               _sz, _vp, ctypes.POINTER(ctypes.c_int64), _vp]),
    "glam_triplet_layer_param_grads_sets": (_i32, [_i32, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i64, _i32, _i32, _i32, _i32, _i32]
                                            + [_vp] * 13 + [_vp, _sz, _vp]),
    "glam_triplet_layer_bwd": (_i32, [_vp] * 14 + [_i64, _i64, _i32, _i32, _i32, _f32] + [_vp] * 4 + [_sz, _vp]),
}

_lib = None


class GlamHipError(RuntimeError):
    pass


def load():
    """Load libglam_hip.so (after torch, so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlamHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C glam_amd/csrc`). glam_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and the header disagree
        fn.restype, fn.argtypes = res, args
    if lib.glam_abi_version() != ABI_VERSION:
        raise GlamHipError(f"{LIB_PATH}: ABI version {lib.glam_abi_version()}, this package binds version {ABI_VERSION} "
                           "(a stale build: `make -C glam_amd/csrc`)")
    _lib = lib
    return lib


def route_enabled(name):
    """Whether the library's alternative route ``name`` ("x3", "wgrad_x3") is on in this process.  The library reads the environment
    (``GLAM_X3``, ``GLAM_WGRAD_X3``) once and with its own parsing; every host-side route decision asks it instead of re-parsing."""
    rc = load().glam_route_enabled(name.encode())
    if rc < 0:
        check(rc, f"glam_route_enabled({name})")
    return rc == 1


def check(rc, what):
    if rc == 0:
        return
    msg = load().glam_last_error().decode("utf-8", "replace")
    raise GlamHipError(f"{what} failed (code {rc}): {msg}")


class kernel_timer:
    """``with kernel_timer() as kt: <eager calls>`` then ``kt.records()`` -> ``[(kernel name, grid blocks, microseconds), ...]`` in
    launch order: per-dispatch begin / end timestamps of every kernel the library launched inside the block (``glam_prof_*``)."""

    def __init__(self, capacity=4096):
        self.capacity, self.n = capacity, 0

    def __enter__(self):
        check(load().glam_prof_begin(self.capacity), "glam_prof_begin")
        return self

    def __exit__(self, *exc):
        self.n = load().glam_prof_end()
        return False

    def records(self):
        lib, out = load(), []
        name, grid, us = ctypes.create_string_buffer(128), ctypes.c_int32(), ctypes.c_float()
        for i in range(self.n):
            check(lib.glam_prof_read(i, name, 128, ctypes.byref(grid), ctypes.byref(us)), "glam_prof_read")
            out.append((name.value.decode(), int(grid.value), float(us.value)))
        return out


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """``hipStream_t`` of torch's current stream on the current device (the raw C accessor when torch exposes it:
    ``torch.cuda.current_stream()`` builds a Python ``Stream`` object and costs ~15 us per call, which an eager
    step pays four times)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(*tensors):
    """Every op argument must already live in HBM; nothing is silently copied or computed on CPU."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GlamHipError("glam_amd ops run on an MI355X HIP device only (got a CPU tensor); "
                               "there is no CPU fallback — move the batch and the model to 'cuda'.")
        if dev is None:
            dev = t.device
            # the kernels are enqueued on the CURRENT device's stream: one process per GPU (torch.cuda.set_device(local_rank)),
            # never silently on another device's stream
            if dev.index is not None and dev.index != torch.cuda.current_device():
                raise GlamHipError(f"tensors live on {dev} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                                   "call torch.cuda.set_device(...) (one process per GPU)")
        elif t.device != dev:
            raise GlamHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def f32c(t, name="tensor"):
    if t.dtype != torch.float32:
        raise GlamHipError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()
