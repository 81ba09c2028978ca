"""ctypes binding of libu2mkd_hip.so (the C ABI declared in include/u2mkd_hip.h).

There is NO fallback: if the library is missing or a call fails this module
raises.  Tensors cross the boundary as raw device pointers + sizes + the
current HIP stream; torch only provides memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (U2MKD_LIB_SUFFIX: a variant library built by tools/build_variant.sh -- same-box A/B runs only)
LIB_PATH = os.path.join(_HERE, 'lib', 'libu2mkd_hip%s.so' % os.environ.get('U2MKD_LIB_SUFFIX', ''))

_i32, _i64, _sz, _f32, _p = C.c_int32, C.c_int64, C.c_size_t, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/u2mkd_hip.h one to one
SIGNATURES = {
    'u2mkd_version': (C.c_int, []),
    'u2mkd_last_error': (C.c_char_p, []),
    'u2mkd_hash': (C.c_int, [_p, _i64, _p, _p]),
    'u2mkd_kernel_hash': (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    'u2mkd_hash_table_bytes': (_sz, [_i64]),
    'u2mkd_hash_table_build': (C.c_int, [_p, _i64, _p, _p]),
    'u2mkd_hash_table_query': (C.c_int, [_p, _i64, _p, _i64, _p, _p]),
    'u2mkd_hash_table_query2': (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p]),
    'u2mkd_kmap_build_table': (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _p]),
    'u2mkd_kmap_invert': (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    'u2mkd_kmap_sizes': (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    'u2mkd_kmap_compact': (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    'u2mkd_downsample_keys': (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    'u2mkd_downsample_keys_checked': (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    'u2mkd_floor_coords': (C.c_int, [_p, _i64, _i32, _p, _p]),
    'u2mkd_unpack_keys': (C.c_int, [_p, _i64, _p, _p]),
    'u2mkd_transpose_weights': (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    'u2mkd_conv_forward': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_conv_forward_sorted': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_debug_conv_forward_sorted': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    'u2mkd_stream_wait_stream': (C.c_int, [_p, _p]),
    'u2mkd_sgd_chunk_elements': (_i32, []),
    'u2mkd_sgd_batch': (C.c_int, [_p, _i32, _i64, _f32, _f32, _f32, _i32, _i32, _p]),
    'u2mkd_mailbox_alloc': (_p, [_sz]),
    'u2mkd_mailbox_free': (C.c_int, [_p]),
    'u2mkd_mailbox_post': (C.c_int, [_p, C.c_uint32, _i32, _p, _i64, _p]),
    'u2mkd_conv_tiles_supported': (_i32, [_i32, _i32, _i32]),
    'u2mkd_conv_tiles_arith': (_i32, [_i32, _i32, _i32]),
    'u2mkd_weight_fragments_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'u2mkd_weight_fragments': (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'u2mkd_weight_fragments_batch': (C.c_int, [_p, _i32, _i64, _p]),
    'u2mkd_conv_forward_tiles': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    'u2mkd_conv_forward_tiles_ep': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_pairs_gather_sum_ep': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_pairs_gather_sum_stats_slab_rows': (_i32, []),
    'u2mkd_pairs_gather_sum_stats_supported': (_i32, [_i32]),
    'u2mkd_pairs_gather_sum_stats': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    'u2mkd_debug_conv_tile_pairs_stamps': (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    'u2mkd_debug_wgrad_stamps': (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    'u2mkd_conv_forward_pairs': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_conv_pairs_x3_supported': (_i32, [_i32, _i32]),
    'u2mkd_conv_forward_pairs_x3': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_linear_forward': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _i32, _p, _p]),
    'u2mkd_bn2d_workspace_bytes': (C.c_size_t, [_i64, _i32, _i64]),
    'u2mkd_bn2d_train_forward': (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _p, C.c_float, C.c_float, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn2d_eval_forward': (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _p, C.c_float, _i32, _p, _p, _p, _p]),
    'u2mkd_bn2d_backward': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn2d_local_stats': (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p]),
    'u2mkd_bn2d_apply': (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_bn2d_backward_local': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn2d_backward_local_keep': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_bn2d_backward_apply': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_linear_forward_x3': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p]),
    'u2mkd_conv_pairs_f16x2_supported': (_i32, [_i32, _i32]),
    'u2mkd_conv_forward_pairs_f16x2': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_linear_forward_f16x2': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p]),
    'u2mkd_pairs_capacity': (_i64, [_i64, _i64, _i32]),
    'u2mkd_pairs_build': (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'u2mkd_pairs_gather_sum': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_kmap_rowmask': (C.c_int, [_p, _i64, _i32, _p, _p]),
    'u2mkd_tile_schedule_workspace_bytes': (_sz, [_i64]),
    'u2mkd_tile_schedule': (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_wgrad_plan_ints': (_i32, [_i32]),
    'u2mkd_wgrad_plan': (C.c_int, [_p, _i32, _i64, _p, _p]),
    'u2mkd_conv_wgrad_pairs_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32]),
    'u2mkd_conv_wgrad_pairs': (C.c_int, [_p, _i32, _p, _i32, _p, _p, _i64, _i32, _i32, _p, _sz, _p, _p]),
    'u2mkd_conv_forward_tiles_bf16': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_conv_wgrad_pairs_bf16': (C.c_int, [_p, _i32, _p, _i32, _p, _p, _i64, _i32, _i32, _p, _sz, _p, _p]),
    'u2mkd_conv_forward_pairs_bf16': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_linear_forward_bf16': (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p]),
    'u2mkd_pairs_gather_sum_bf16': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_convolution_workspace_bytes': (_sz, [_i64, _i64, _i32, _i32, _p, _i32]),
    'u2mkd_convolution_forward': (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _i32, _i32, _p, _sz, _p]),
    'u2mkd_convolution_backward': (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _p, _sz, _p]),
    'u2mkd_bn_num_slabs': (_i64, [_i64]),
    'u2mkd_bn_train_forward': (C.c_int, [_p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_bn_train_forward_counted': (C.c_int, [_p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_bn_eval_forward': (C.c_int, [_p, _i64, _i32, _p, _p, _f32, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_train_forward_res': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_bn_train_forward_from_partial': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_bn_eval_forward_res': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _f32, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_res': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn_backward': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_bn_local_stats': (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    'u2mkd_bn_merge_stats': (C.c_int, [_p, _i32, _i32, _f32, _f32, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn_merge_stats_counted': (C.c_int, [_p, _i32, _i32, _f32, _f32, _p, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn_backward_local_keep': (C.c_int, [_p, _p, _p, _i32, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_bn_apply': (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_bn_backward_local': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_apply': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_train_forward_res_bf16': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_bn_eval_forward_res_bf16': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _f32, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_res_bf16': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'u2mkd_bn_local_stats_bf16': (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    'u2mkd_bn_apply_res': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_bn_backward_local_res': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_apply_res': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_bn_apply_res_bf16': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_bn_backward_local_res_bf16': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_apply_res_bf16': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    'u2mkd_bn_apply_bf16': (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_bn_backward_local_bf16': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_bn_backward_apply_bf16': (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p]),
    'u2mkd_sptr_window_keys': (C.c_int, [_p, _p, _i64, _p, _p, _f32, _f32, _f32, _p, _p]),
    'u2mkd_select_mse_partials': (_i32, []),
    'u2mkd_select_mse_forward': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    'u2mkd_select_mse_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    'u2mkd_sptr_plan_prepare_workspace_bytes': (_sz, []),
    'u2mkd_sptr_plan_prepare': (C.c_int, [_p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _p, _p, _p, _p, _p, _p]),
    'u2mkd_sptr_window_ranges': (C.c_int, [_p, _i64, _p, _p, _p]),
    'u2mkd_sptr_quant_coords': (C.c_int, [_p, _p, _i64, _p, _f32, _f32, _f32, _f32, _f32, _f32, _p, _p, _p]),
    'u2mkd_sptr_attention_forward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _f32, _i64, _i32, _i32, _p, _p, _p]),
    'u2mkd_sptr_backward_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'u2mkd_sptr_attention_backward': (C.c_int, [_p] * 14 + [_i32, _i32, _f32, _i32, _i64, _i32, _i32] + [_p, _p, _sz] + [_p] * 7),
    'u2mkd_sptr_tiles_workspace_bytes': (_sz, [_i64, _i32]),
    'u2mkd_sptr_attention_forward_tiles': (C.c_int, [_p, _p, _p, _i64, _f32] + [_p] * 8 + [_i32, _i32, _f32, _i64, _i32, _i32, _p, _i64, _p, _p, _sz, _p]),
    'u2mkd_sptr_attention_forward_strided': (C.c_int, [_p, _p, _p, _i64, _f32] + [_p] * 8 + [_i32, _i32, _f32, _i64, _i32, _i32, _p, _i64, _p, _p]),
    'u2mkd_sptr_attention_backward_strided': (C.c_int, [_p, _p, _p, _i64, _f32, _p, _p, _i64] + [_p] * 9 + [_i32, _i32, _f32, _i32, _i64, _i32, _i32]
                                              + [_p, _p, _sz] + [_p, _p, _p, _i64, _p, _p, _p, _p]),
    'u2mkd_sptr_table_reduce': (C.c_int, [_p, _i64, _i32, _i32, _f32, _p, _p, _p, _p]),
    # the ten sptr_cuda functions, argument for argument (csrc/sptr_ops.hip)
    'u2mkd_sptr_precompute_all': (C.c_int, [_i32, _i32, C.c_uint32] + [_p] * 8),
    'u2mkd_sptr_attention_step1_forward': (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.c_uint32] + [_p] * 6),
    'u2mkd_sptr_attention_step1_backward': (C.c_int, [_i32, _i32, _i32, _i32, C.c_uint32] + [_p] * 10),
    'u2mkd_sptr_attention_step2_forward': (C.c_int, [_i32] * 5 + [_p] * 6),
    'u2mkd_sptr_attention_step2_backward': (C.c_int, [_i32] * 5 + [_p] * 10),
    'u2mkd_sptr_dot_prod_with_idx_forward': (C.c_int, [_i32] * 6 + [_p] * 10),
    'u2mkd_sptr_dot_prod_with_idx_all_forward': (C.c_int, [_i32] * 6 + [_p] * 10),
    'u2mkd_sptr_dot_prod_with_idx_backward': (C.c_int, [_i32] * 6 + [_p] * 14),
    'u2mkd_sptr_attention_step2_with_rel_pos_value_forward': (C.c_int, [_i32] * 5 + [_p] * 8),
    'u2mkd_sptr_attention_step2_with_rel_pos_value_backward': (C.c_int, [_i32] * 6 + [_p] * 13),
    'u2mkd_count': (C.c_int, [_p, _i64, _p, _i64, _p]),
    'u2mkd_voxelize_forward': (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    'u2mkd_voxelize_backward': (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    'u2mkd_devoxelize_forward': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_devoxelize_backward': (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    'u2mkd_segment_sum': (C.c_int, [_p, _i32, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_voxelize_backward_bf16': (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, _p]),
    'u2mkd_devoxelize_forward_bf16': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_segment_sum_bf16': (C.c_int, [_p, _i32, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_c2l_plan': (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _i32, _p, _p, _p]),
    'u2mkd_l2c_keys': (C.c_int, [_p, _p, _i32, _i64, _i32, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _p]),
    'u2mkd_l2c_finish': (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p]),
    'u2mkd_up_bilinear_forward': (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _p, _p]),
    'u2mkd_up_bilinear_backward': (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_maxpool3s2_forward': (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p]),
    'u2mkd_maxpool3s2_backward': (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    'u2mkd_transpose_batched': (C.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    'u2mkd_transpose_batched_scaled': (C.c_int, [_p, _p, _i32, _i32, _i32, C.c_float, _p]),
    'u2mkd_l2c_combine_forward': (C.c_int, [_p, _p, _p, _p] + [_i32] * 11 + [_p, _p]),
    'u2mkd_l2c_combine_backward': (C.c_int, [_p] + [_i32] * 6 + [_p, _p]),
    'u2mkd_up_plan': (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _p, _p, _p]),
    'u2mkd_upbn_stats': (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_upbn_dense_grad': (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p]),
    'u2mkd_csr_workspace_bytes': (_sz, [_i64, _i64]),
    'u2mkd_lovasz_errors': (C.c_int, [_p, _p, _i32, _i64, _i32, _p, _p, _p]),
    'u2mkd_lovasz_gather': (C.c_int, [_p, _p, _i32, _i64, _i32, _p, _p]),
    'u2mkd_lovasz_partials': (_i64, [_i64, _i32]),
    'u2mkd_lovasz_terms': (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    'u2mkd_ce_partials': (_i64, [_i64]),
    'u2mkd_ce_forward': (C.c_int, [_p, _p, _i32, _i64, _i32, _p, _p, _p, _p]),
    'u2mkd_ce_backward': (C.c_int, [_p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p]),
    'u2mkd_kl_forward': (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    'u2mkd_kl_backward': (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    'u2mkd_lovasz_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p]),
    'u2mkd_csr_build': (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p]),
    'u2mkd_devoxelize_plan_workspace_bytes': (C.c_size_t, [_i64, _i64]),
    'u2mkd_devoxelize_plan': (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    'u2mkd_ti_weights': (C.c_int, [_p, _p, _i64, _f32, _p, _p, _p]),
}

_lib = None
_raw_stream = torch._C._cuda_getCurrentRawStream
_current_device = torch._C._cuda_getDevice


def load():
    """Load the shared library (once) and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -m u2mkd_amd.build` '
            '(u2mkd_amd has no CPU or PyTorch fallback path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def stream() -> int:
    """Raw hipStream_t of torch's current stream (the fast private accessor: this runs ~700 times a step)."""
    return _raw_stream(_current_device())


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


_fns = {}


def call(name: str, *args):
    """Call an int-returning entry point; raise RuntimeError on failure."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        msg = load().u2mkd_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{name} failed (rc={rc}): {msg}')


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('u2mkd_amd operators run on the HIP device only (got a CPU tensor); '
                               'there is no CPU fallback')
