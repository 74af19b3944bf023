"""ctypes binding of libsbev_hip.so (C ABI: include/sbev_hip.h).

The library is the product: if it is missing or fails to load, importing the operators raises --
there is no PyTorch / CPU fallback anywhere in this package (the reference silently falls back to
``F.grid_sample``, models/csrc/wrapper.py:4-11; we deliberately do not).
"""
import ctypes
import os

# Load order matters: the PyTorch-ROCm wheel bundles its own libamdhip64.so.  Importing torch FIRST makes
# the dynamic linker satisfy libsbev_hip.so's NEEDED libamdhip64.so.7 with that already-loaded runtime, so
# our kernels and torch's streams / allocations live in ONE HIP runtime.  (Loaded the other way round the
# process ends up with two runtimes and every launch on a torch stream fails with "no ROCm-capable device".)
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SBEV_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libsbev_hip.so')   # override: A/B builds

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/sbev_hip.h one to one
SIGNATURES = {
    'sbev_abi_version': (ctypes.c_int, []),
    'sbev_set_box_convention': (ctypes.c_int, [ctypes.c_int]),
    'sbev_get_box_convention': (ctypes.c_int, []),
    'sbev_last_error': (ctypes.c_char_p, []),
    'sbev_device_count': (ctypes.c_int, []),
    'sbev_msmv_fwd': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64,
                                     _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_project_select': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           _vp, _vp, _vp, _vp, _vp]),
    'sbev_sampling_front': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double),
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           _vp, _vp, _vp]),
    'sbev_linear_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_linear_splitk_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_splitk_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp,
                                              ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_layer_norm_f32': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int, _vp]),
    'sbev_ln_linear_f32': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int, _vp]),
    'sbev_adaptive_mixing_f32': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_float, _vp]),
    'sbev_sasa_f32': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double), _vp, _vp,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_refine_bbox': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_nchw_to_nhwc_f32': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_nchw_to_nhwc_f32_indirect': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_nchw_to_nhwc_f32_multi_indirect': (ctypes.c_int, [_vp, ctypes.c_int, _c_i32p, ctypes.POINTER(_vp), ctypes.c_int64, ctypes.c_int, _c_i32p, _vp]),
    'sbev_nchw_to_nhwc_b16': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_nchw_to_nhwc_b16_indirect': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_copy_indirect': (ctypes.c_int, [_vp, ctypes.c_int, _c_i32p, ctypes.POINTER(_vp), _c_i64p, _vp]),
    'sbev_finish_outputs': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp]),
    'sbev_finish_outputs_indirect': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp]),
    'sbev_linear3_ln_relu_f32': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, ctypes.c_float, _vp,
                                                ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_decoder_workspace_bytes': (ctypes.c_int64, [_vp]),
    'sbev_decoder_launches_per_layer': (ctypes.c_int, [_vp, _vp]),
    'sbev_decoder_forward': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, ctypes.c_int64, _vp]),
    'sbev_decoder_forward_lazy': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                 _vp, ctypes.c_int64, _vp]),
    'sbev_decoder_lazy_supported': (ctypes.c_int, [_vp]),
    'sbev_decoder_lazy_scan_launch': (ctypes.c_int, [ctypes.c_int]),
    'sbev_lazy_relayout_tiles': (ctypes.c_int64, [ctypes.c_int, _c_i32p, ctypes.c_int64, ctypes.c_int]),
    'sbev_sample_and_project_touch': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp,
                                                     ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _c_i32p, _vp, _vp]),
    'sbev_nchw_to_nhwc_lazy': (ctypes.c_int, [_vp, _c_i32p, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, _c_i32p, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_init': (ctypes.c_int, []),
    'sbev_decoder_out_fold': (ctypes.c_int, [ctypes.c_int]),
    'sbev_linear_out8_min_rows': (ctypes.c_int, [ctypes.c_int]),
    'sbev_debug_out_fold_drop': (ctypes.c_int, [ctypes.c_int]),
    'sbev_profile_sampler': (ctypes.c_int, [ctypes.c_int]),
    'sbev_profile_stride': (ctypes.c_int, [ctypes.c_int]),
    'sbev_profile_sampler_read': (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    'sbev_splitk_reduce_f32': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_split_bf16x3_weights': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_linear_bf16x3_strip_ok': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_bf16x3_strip': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_int, _vp]),
    'sbev_linear_bf16x3': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_linear_splitk_bf16x3': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_bf16s_image_elems': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_split_bf16s_rows': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_pack_bf16s_frags': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_linear_gen_weight_stationary': (ctypes.c_int, [ctypes.c_int]),
    'sbev_linear_bf16s_gen_ok': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_bf16s_gen': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_decoder_mixed_up_log2': (ctypes.c_int, [_vp]),
    'sbev_pack_f16s_frags': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_linear_f16s_gen': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_f16s_pairs': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_f16s_tensor_scale': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    'sbev_gemm_tn_f16s_ok': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    'sbev_gemm_tn_f16s': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_adaptive_mixing_bwd_max_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_float, _vp]),
    'sbev_adaptive_mixing_pairs_f16': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_float, ctypes.c_int, _vp]),
    'sbev_linear_splitk_f16s': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_f16s_out_scale': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    'sbev_linear_splitk_f16s_xdev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_linear_bf16s_out_ok': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_bf16s_out_plan': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_splitk_bf16s': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_msmv_bwd': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _c_i32p, ctypes.c_int,
                                     ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    'sbev_msmv_fwd_ring': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64,
                                          _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int, _vp]),
    'sbev_sample_and_project': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp,
                                               ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    'sbev_decoder_capture': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    'sbev_capture_begin': (ctypes.c_int, [_vp]),
    'sbev_capture_end': (ctypes.c_int, [_vp, _vp]),
    'sbev_graph_launch': (ctypes.c_int, [_vp, _vp]),
    'sbev_graph_num_nodes': (ctypes.c_int64, [_vp]),
    'sbev_graph_destroy': (ctypes.c_int, [_vp]),
    'sbev_head_prepare': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_head_denorm': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), _vp, ctypes.c_int64, _vp]),
    'sbev_nms_free_decode': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    'sbev_profile_read': (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int]),
    'sbev_linear_splitk_plan': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sbev_linear_group_f32': (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    'sbev_copy_widen_f32': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int64, _vp]),
    'sbev_msmv_buffer_taps': (ctypes.c_int, [ctypes.c_int]),
    'sbev_sample_mix_supported': (ctypes.c_int, [ctypes.c_int] * 6),
    'sbev_sample_mix_slabs_ok': (ctypes.c_int, [_c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64p, ctypes.c_int64]),
    'sbev_sample_mix_f32': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64, _vp, _vp, _c_i32p, ctypes.c_int,
                                           _vp, _vp, ctypes.c_int, ctypes.c_float, _vp]),
    'sbev_sample_mix_pairs_f16': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64, _vp, _vp, _c_i32p, ctypes.c_int,
                                                 _vp, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp]),
    'sbev_decoder_fuse_sample_mix': (ctypes.c_int, [ctypes.c_int]),
    'sbev_query_order_max': (ctypes.c_int, []),
    'sbev_query_order': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'sbev_sample_mix_f32_ordered': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64, _vp, _vp, _c_i32p, ctypes.c_int,
                                                   _vp, _vp, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    'sbev_sample_mix_pairs_f16_ordered': (ctypes.c_int, [ctypes.POINTER(_vp), _c_i32p, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64, _vp, _vp, _c_i32p, ctypes.c_int,
                                                         _vp, _vp, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp, _vp]),
    'sbev_decoder_query_order': (ctypes.c_int, [ctypes.c_int]),
    'sbev_decoder_chain_pack_floats': (ctypes.c_int64, [_vp]),
    'sbev_decoder_chain_pack': (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    'sbev_decoder_row_chain': (ctypes.c_int, [ctypes.c_int]),
    'sbev_decoder_chain_pair': (ctypes.c_int, [ctypes.c_int]),
    'sbev_decoder_chain_pair_timeouts': (ctypes.c_int64, []),
    'sbev_decoder_chain_pair_faults': (ctypes.c_int64, []),
    'sbev_decoder_chain_pair_faults_ack': (ctypes.c_int64, []),
    'sbev_debug_chain_pair_drop': (ctypes.c_int, [ctypes.c_int]),
    'sbev_gemm_f32_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    'sbev_gemm_f32': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    'sbev_colsum_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int]),
    'sbev_layer_norm_bwd_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int]),
    'sbev_gemm_f32_multi_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    'sbev_gemm_f32_multi': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int64, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_int, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    'sbev_layer_norm_bwd_rows': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, _vp, ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_layer_norm_param_group': (ctypes.c_int, [ctypes.POINTER(_vp)] * 7 + [_c_i32p] * 4 + [ctypes.c_int, ctypes.c_int64, _vp]),
    'sbev_colsum_group': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _c_i32p, _c_i32p, _c_i32p, ctypes.c_int, ctypes.c_int64, _vp]),
    'sbev_bias_relu_bwd_acc': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, _vp]),
    'sbev_layer_norm_bwd_acc': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, _vp, _vp, _vp,
                                               ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_bias_relu_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, _vp, _vp]),
    'sbev_layer_norm_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, _vp, _vp, _vp,
                                           ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_linear3_ln_relu_ex_f32': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp,
                                                   ctypes.c_int64, ctypes.c_int, _vp]),
    'sbev_adaptive_mixing_bwd_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_float, _vp]),
    'sbev_sasa_train_fwd_f32': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double), _vp, _vp,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, _vp]),
    'sbev_sasa_bwd_f32': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp, _vp, _vp,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, _vp]),
    'sbev_msmv_bwd_ex': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _c_i32p, ctypes.c_int,
                                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _c_i64p, ctypes.c_int64, _c_i64p, ctypes.c_int64,
                                        _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    'sbev_project_select_bwd': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    'sbev_sampling_front_bwd': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_double),
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    'sbev_refine_bbox_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    'sbev_dropout_f32': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, _vp]),
    'sbev_sasa_train_fwd_f32_ds': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double), _vp, _vp,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, _vp, _vp]),
    'sbev_sasa_bwd_f32_ds': (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp, _vp, _vp,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, _vp, _vp]),
    'sbev_dropout_f32_ds': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_uint64, _vp, ctypes.c_float, _vp]),
}

_lib = None


class SbevError(RuntimeError):
    """A libsbev_hip.so entry point returned a negative status (message from sbev_last_error())."""


def load():
    """dlopen the HIP library (once) and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not a fallback: the only alternative to a prebuilt library is building the same HIP sources now
        try:
            from .csrc import build as _build
            _build.build()
        except Exception as e:      # noqa: BLE001
            raise ImportError(
                'sparsebev_amd: %s not found and building it failed (%s). Build it with `python -m sparsebev_amd.csrc.build` '
                '(needs hipcc, targets gfx950). There is no CPU / PyTorch fallback for this path.' % (LIB_PATH, e))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale: loud on purpose
        fn.restype = res
        fn.argtypes = args
    if lib.sbev_abi_version() != 1:
        raise ImportError('sparsebev_amd: libsbev_hip.so ABI %d != 1 (stale build?)' % lib.sbev_abi_version())
    _lib = lib
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        # per-device setup that must not run inside somebody's stream capture (pair-mode fault word); retried lazily on failure.  Only when
        # the process already has its device context: a rank of a multi-GPU job that imports this before torch.cuda.set_device(LOCAL_RANK)
        # must not open a context on GPU 0 for it (the install then happens at its first sbev_decoder_workspace_bytes, on ITS device)
        lib.sbev_init()
    return lib


class PairFaultError(SbevError):
    """SBEV_EFAULT: an EARLIER decoder step lost a pair-mode hand-off (the GPU was shared or preempted for about a second) -- that
    step's outputs are invalid.  By the time this is raised pair mode is off and the fault acknowledged: repeat the step(s) since
    the last result you synchronised on and verified with ``runtime.check_pair_faults()``."""


EFAULT = -4


def check(status, what):
    if status == EFAULT:
        from . import runtime
        msg = load().sbev_last_error().decode('utf-8', 'replace')
        runtime._on_pair_fault()
        raise PairFaultError('%s refused (%d): %s' % (what, status, msg))
    if status != 0:
        msg = load().sbev_last_error().decode('utf-8', 'replace')
        raise SbevError('%s failed (%d): %s' % (what, status, msg))
