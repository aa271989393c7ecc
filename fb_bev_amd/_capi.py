"""ctypes binding of libfbbev_hip.so (include/fbbev.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; the kernels are reached only
through the C ABI with raw pointers.  There is NO CPU fallback: if the library is missing or a
tensor is not on a GPU the call raises.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FBBEV_LIB: alternate build of the same C ABI (debug / tuning variants); there is still no non-HIP path
LIB_PATH = os.environ.get('FBBEV_LIB') or os.path.join(_HERE, 'libfbbev_hip.so')

# name -> (restype, argtypes); mirrors include/fbbev.h one to one
SIGNATURES = {
    'fbbev_version': (c_int, []),
    'fbbev_bev_pool_v2_fwd': (c_int, [c_int, c_int] + [c_void_p] * 8 + [c_void_p]),
    'fbbev_bev_pool_v2_bwd': (c_int, [c_int, c_int] + [c_void_p] * 10 + [c_void_p]),
    'fbbev_lidar_coor': (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p, c_void_p]),
    'fbbev_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'fbbev_tokens_from_nchw_levels': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'fbbev_tokens_from_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    'fbbev_tokens_from_nchw_pos': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    'fbbev_rank_workspace_bytes': (c_size_t, [c_int64]),
    'fbbev_rank_build': (c_int, [c_void_p] + [c_int] * 5 + [c_void_p] * 3 + [c_void_p] * 7 +
                         [c_void_p, c_size_t, c_void_p]),
    'fbbev_rank_build_depth': (c_int, [c_void_p, c_void_p, c_float] + [c_int] * 5 + [c_void_p] * 3 + [c_void_p] * 7 +
                               [c_void_p, c_size_t, c_void_p]),
    'fbbev_lift_rank_build': (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p] * 3 + [c_void_p] * 7 +
                              [c_void_p, c_size_t, c_void_p]),
    'fbbev_cam_key_words': (c_size_t, [c_int] * 2),
    'fbbev_lift_rank_build_cached': (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p] * 3 + [c_void_p] * 7 +
                                     [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'fbbev_pool_tile_index_cached': (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'fbbev_lift_splat_fused_ws_bytes': (c_size_t, [c_int] * 9),
    'fbbev_lift_splat_fused_ws_offsets': (c_int, [c_int] * 9 + [c_void_p]),
    'fbbev_lift_splat_fused': (c_int, [c_void_p] * 12 + [c_int] * 6 + [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_int64, c_int64, c_int,
                                       c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'fbbev_pool_dense_workspace_bytes': (c_size_t, [c_int] * 4),
    'fbbev_pool_tile_index': (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    'fbbev_bev_pool_v2_dense_fwd': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_int64, c_int64, c_void_p,
                                            c_size_t, c_int, c_int, c_void_p]),
    'fbbev_diag_pool_store_floor': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                            c_void_p]),
    'fbbev_bev_pool_v2_dense_fwd_add': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_int64, c_int64, c_void_p,
                                                c_size_t, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_pool_zmean_split': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_size_t,
                                       c_void_p]),
    'fbbev_pool_zmean': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'fbbev_pool_zmean_rows': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'fbbev_pool_dense_bwd_workspace_bytes': (c_size_t, [c_int] * 9),
    'fbbev_bev_pool_v2_dense_bwd': (c_int, [c_void_p, c_int64, c_int64] + [c_void_p] * 6 + [c_int] * 10 +
                                    [c_void_p] * 3 + [c_size_t, c_void_p]),
    'fbbev_bev_pool_v2_dense_bwd_z': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_float] + [c_void_p] * 6 + [c_int] * 10 +
                                    [c_void_p] * 3 + [c_size_t, c_void_p]),
    'fbbev_history_flow': (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p]),
    'fbbev_history_warp': (c_int, [c_void_p, c_int64, c_void_p] + [c_int] * 5 + [c_void_p, c_int64, c_void_p]),
    'fbbev_history_warp_e': (c_int, [c_void_p, c_int64, c_void_p] + [c_int] * 5 + [c_void_p, c_int64, c_int, c_void_p]),
    'fbbev_layernorm': (c_int, [c_void_p] * 4 + [c_float, c_int64, c_int, c_void_p, c_void_p]),
    'fbbev_rows_linear_x3_fragment_bytes': (c_size_t, [c_int, c_int]),
    'fbbev_rows_linear_x3_fragments': (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'fbbev_rows_linear_x3': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    'fbbev_rows_linear_x3_add': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                         c_void_p, c_int64, c_void_p]),
    'fbbev_rows_linear_x3_ln': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                        c_float, c_void_p, c_int64, c_void_p]),
    'fbbev_rows_ffn_x3': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    'fbbev_rows_tail_ffn_x3': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    'fbbev_rows_tail_ffn_x3_planes': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_int64, c_void_p, c_void_p]),
    'fbbev_rows_linear_x3_planes': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_rows_linear_x3_planes_e': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fused_supported': (c_int, [c_int] * 10),
    'fbbev_da_cross_attn_fused_e': (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 4 + [c_int] * 10 +
                                    [c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fused': (c_int, [c_void_p] * 8 + [c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 4 + [c_int] * 10 +
                                  [c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fused_ln': (c_int, [c_void_p] * 8 + [c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 4 +
                                     [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float] + [c_int] * 10 +
                                     [c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_rows_to_head_planes': (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_msda_self_fused_supported': (c_int, [c_int] * 8),
    'fbbev_msda_self_fused': (c_int, [c_void_p] * 3 + [c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 4 + [c_int] * 10 +
                              [c_void_p, c_void_p]),
    'fbbev_msda_self_fused_ln': (c_int, [c_void_p] * 3 + [c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 4 +
                                 [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float] + [c_int] * 10 + [c_void_p, c_void_p]),
    'fbbev_rows_wgrad_x3_ws_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'fbbev_rows_wgrad_x3': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    'fbbev_rows_linear_x3_train': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                           c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'fbbev_sum_leading': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    'fbbev_sum_partials': (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    'fbbev_diag_fill': (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    'fbbev_touch': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'fbbev_softmax_groups': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'fbbev_softmax_groups_bwd': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'fbbev_layernorm_bwd_partials': (c_int, [c_int64]),
    'fbbev_layernorm_bwd': (c_int, [c_void_p] * 3 + [c_float, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'fbbev_history_conv': (c_int, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    'fbbev_history_conv_e': (c_int, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_history_conv_bf16': (c_int, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'fbbev_history_conv_bf16x3': (c_int, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_history_fused_x3_vm': (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_history_step_x3_vm': (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'fbbev_history_conv_vm': (c_int, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_history_fused_vm': (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 5 + [c_int] * 7 +
                               [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_history_warp_vm': (c_int, [c_void_p, c_int64, c_void_p] + [c_int] * 6 + [c_void_p, c_int64, c_int, c_void_p]),
    'fbbev_history_frame_vm': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    'fbbev_conv3d_ndhwc': (c_int, [c_void_p] * 4 + [c_int] * 14 + [c_void_p, c_void_p]),
    'fbbev_conv2d_nhwc': (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p, c_void_p]),
    'fbbev_conv3d_ndhwc_bf16': (c_int, [c_void_p] * 4 + [c_int] * 15 + [c_void_p, c_void_p]),
    'fbbev_conv3d_k3s1_tiled_bf16': (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_void_p]),
    'fbbev_conv3d_dgrad_ndhwc': (c_int, [c_void_p] * 3 + [c_int] * 12 + [c_void_p, c_void_p]),
    'fbbev_conv3d_wgrad_ndhwc': (c_int, [c_void_p] * 2 + [c_int] * 12 + [c_void_p, c_void_p]),
    'fbbev_blend_levels_ndhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    'fbbev_msda_fwd': (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p]),
    'fbbev_point_sampling': (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_float, c_float] + [c_void_p] * 4),
    'fbbev_da_cross_attn_fwd': (c_int, [c_void_p] * 9 + [c_int] * 10 + [c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fwd_zt_fuses_softmax': (c_int, [c_int] * 11),
    'fbbev_da_cross_attn_fwd_zt': (c_int, [c_void_p] * 9 + [c_int] * 10 + [c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fwd_e': (c_int, [c_void_p] * 9 + [c_int] * 10 + [c_float, c_float, c_int, c_int, c_int] + [c_void_p, c_void_p]),
    'fbbev_da_cross_attn_bwd': (c_int, [c_void_p] * 10 + [c_int] * 10 + [c_float, c_float, c_int, c_int] + [c_void_p] * 5),
    'fbbev_da_cross_attn_bwd_ws_bytes': (c_size_t, [c_int] * 9 + [c_void_p]),
    'fbbev_da_cross_attn_bwd_ws_bytes_za': (c_size_t, [c_int] * 10 + [c_void_p]),
    'fbbev_da_cross_attn_bwd_ws': (c_int, [c_void_p] * 10 + [c_int] * 10 + [c_float, c_float, c_int, c_int] + [c_void_p] * 4 +
                                   [c_void_p, c_void_p, c_size_t, c_void_p]),
    'fbbev_da_cross_attn_bwd_ws_grid': (c_int, [c_void_p] * 10 + [c_int] * 10 + [c_float, c_float, c_int, c_int] + [c_void_p] * 4 +
                                        [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_da_cross_attn_bwd_planes': (c_int, [c_void_p] * 10 + [c_int] * 10 + [c_float, c_float, c_int, c_int] + [c_void_p] * 4 +
                                       [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fbbev_da_cross_attn_bwd_planes_supported': (c_int, [c_int] * 10 + [c_void_p, c_int]),
    'fbbev_msda_fwd_fused': (c_int, [c_void_p] * 6 + [c_int] * 9 + [c_void_p, c_void_p]),
    'fbbev_msda_bwd': (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_void_p]),
    'fbbev_msda_bwd_ws_bytes': (c_size_t, [c_int] * 7 + [c_void_p]),
    'fbbev_msda_bwd_ws': (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    'fbbev_volume_zreduce': (c_int, [c_void_p, c_int64, c_int, c_int64, c_float, c_void_p, c_void_p]),
    'fbbev_volume_zreduce_inner': (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    'fbbev_volume_z_to_front': (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    'fbbev_value_rows_to_head_planes': (c_int, [c_void_p, c_int64] + [c_int] * 5 + [c_void_p, c_void_p]),
    'fbbev_da_cross_attn_fwd_planes_supported': (c_int, [c_int] * 9),
    'fbbev_da_cross_attn_fwd_planes': (c_int, [c_void_p] * 9 + [c_int] * 10 + [c_float, c_float, c_int, c_int, c_int] + [c_void_p, c_void_p]),
}

_lib = None


class FbbevError(RuntimeError):
    pass


def declare(cdll):
    """Attach restype/argtypes for every symbol of the ABI (raises if one is missing)."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return cdll


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FbbevError(
                f'{LIB_PATH} not found: build it with `python -m fb_bev_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        _lib = declare(ctypes.CDLL(LIB_PATH))
    return _lib


def _check(code, what):
    if code != 0:
        kind = 'invalid argument' if code < 0 else 'hipError_t'
        raise FbbevError(f'{what} failed: {kind} {code}')


def _dev(t, dtype, name, contiguous=True):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FbbevError(f'{name} must be a GPU tensor (no CPU fallback in fb_bev_amd)')
    if t.dtype != dtype:
        raise FbbevError(f'{name} must be {dtype}, got {t.dtype}')
    if contiguous and not t.is_contiguous():
        raise FbbevError(f'{name} must be contiguous')
    return c_void_p(t.data_ptr())


def require_gpu(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FbbevError(f'{name} must be a GPU tensor (no CPU fallback in fb_bev_amd)')


def _on(t):
    """Device guard for the tensor's GPU (the OptionalCUDAGuard of bev_pool.cpp:40,86)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FbbevError('expected a GPU tensor (no CPU fallback in fb_bev_amd)')
    return torch.cuda.device(t.device)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


F32, I32, I64 = torch.float32, torch.int32, torch.int64


def bev_pool_v2_fwd(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                    interval_lengths):
    c = feat.shape[-1]
    n = interval_starts.numel()
    with _on(depth):
        _check(lib().fbbev_bev_pool_v2_fwd(
            c, n, _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'),
            _dev(ranks_depth, I32, 'ranks_depth'), _dev(ranks_feat, I32, 'ranks_feat'),
            _dev(ranks_bev, I32, 'ranks_bev'), _dev(interval_starts, I32, 'interval_starts'),
            _dev(interval_lengths, I32, 'interval_lengths'), _dev(out, F32, 'out'), _stream()),
            'fbbev_bev_pool_v2_fwd')


def bev_pool_v2_bwd(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                    ranks_bev, interval_starts, interval_lengths):
    c = out_grad.shape[-1]
    n = interval_starts.numel()
    with _on(out_grad):
        _check(lib().fbbev_bev_pool_v2_bwd(
            c, n, _dev(out_grad, F32, 'out_grad'), _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'),
            _dev(ranks_depth, I32, 'ranks_depth'), _dev(ranks_feat, I32, 'ranks_feat'),
            _dev(ranks_bev, I32, 'ranks_bev'), _dev(interval_starts, I32, 'interval_starts'),
            _dev(interval_lengths, I32, 'interval_lengths'), _dev(depth_grad, F32, 'depth_grad'),
            _dev(feat_grad, F32, 'feat_grad'), _stream()), 'fbbev_bev_pool_v2_bwd')


def lidar_coor(xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, coor):
    B, N = trans.shape[:2]
    D, H, W = ds.numel(), ys.numel(), xs.numel()
    with _on(coor):
        _check(lib().fbbev_lidar_coor(
            _dev(xs, F32, 'xs'), _dev(ys, F32, 'ys'), _dev(ds, F32, 'ds'), _dev(rots, F32, 'rots'),
            _dev(trans, F32, 'trans'), _dev(intrins, F32, 'intrins'), _dev(post_rots, F32, 'post_rots'),
            _dev(post_trans, F32, 'post_trans'), _dev(bda, F32, 'bda'), B, N, D, H, W,
            _dev(coor, F32, 'coor'), _stream()), 'fbbev_lidar_coor')


def rank_workspace_bytes(n_points):
    return int(lib().fbbev_rank_workspace_bytes(int(n_points)))


def rank_build(coor, lower3, interval3, grid_size3, ranks_bev, ranks_depth, ranks_feat,
               interval_starts, interval_lengths, interval_rank, counts, workspace, depth=None, depth_threshold=0.01):
    """coor (B,N,D,H,W,3) f32 GPU; lower3/interval3/grid_size3: 3 python floats each (fp32 values).
    depth (B,N,D,H,W): optional BEVDet-era filter, points with depth <= depth_threshold are dropped."""
    B, N, D, H, W, three = coor.shape
    assert three == 3
    arr = ctypes.c_float * 3
    lo, it, gs = arr(*lower3), arr(*interval3), arr(*grid_size3)
    tail = (B, N, D, H, W, ctypes.cast(lo, c_void_p), ctypes.cast(it, c_void_p),
            ctypes.cast(gs, c_void_p), _dev(ranks_bev, I32, 'ranks_bev'),
            _dev(ranks_depth, I32, 'ranks_depth'), _dev(ranks_feat, I32, 'ranks_feat'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
            _dev(interval_rank, I32, 'interval_rank') if interval_rank is not None else c_void_p(0),
            _dev(counts, I32, 'counts'), c_void_p(workspace.data_ptr()),
            workspace.numel() * workspace.element_size())
    with _on(coor):
        if depth is None:
            _check(lib().fbbev_rank_build(_dev(coor, F32, 'coor'), *tail, _stream()), 'fbbev_rank_build')
        else:
            if depth.numel() != B * N * D * H * W:
                raise FbbevError('depth must have one element per frustum point')
            _check(lib().fbbev_rank_build_depth(_dev(coor, F32, 'coor'), _dev(depth, F32, 'depth'), float(depth_threshold),
                                                *tail, _stream()), 'fbbev_rank_build_depth')


def lift_rank_build(xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, lower3, interval3, grid_size3,
                    ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths, interval_rank, counts,
                    workspace, frustum=None, cam_key=None, cache_state=None):
    """Geometry + ranking in one call (no coor tensor); arguments as lidar_coor + rank_build.
    cam_key (int32[cam_key_words(B,N)], initialised to -1) + cache_state (int32[2]): the camera-keyed cache -- the build
    is skipped on the device when the six camera tensors equal the cached key (fbbev_lift_rank_build_cached)."""
    B, N = trans.shape[:2]
    D, H, W = ds.numel(), ys.numel(), xs.numel()
    arr = ctypes.c_float * 3
    lo, it, gs = arr(*lower3), arr(*interval3), arr(*grid_size3)
    with _on(ranks_bev):
        args = [
            _dev(frustum, F32, 'frustum') if frustum is not None else c_void_p(0), _dev(xs, F32, 'xs'), _dev(ys, F32, 'ys'), _dev(ds, F32, 'ds'), _dev(rots, F32, 'rots'),
            _dev(trans, F32, 'trans'), _dev(intrins, F32, 'intrins'), _dev(post_rots, F32, 'post_rots'),
            _dev(post_trans, F32, 'post_trans'), _dev(bda, F32, 'bda'), B, N, D, H, W,
            ctypes.cast(lo, c_void_p), ctypes.cast(it, c_void_p), ctypes.cast(gs, c_void_p),
            _dev(ranks_bev, I32, 'ranks_bev'), _dev(ranks_depth, I32, 'ranks_depth'),
            _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_starts, I32, 'interval_starts'),
            _dev(interval_lengths, I32, 'interval_lengths'),
            _dev(interval_rank, I32, 'interval_rank') if interval_rank is not None else c_void_p(0),
            _dev(counts, I32, 'counts'), c_void_p(workspace.data_ptr()),
            workspace.numel() * workspace.element_size()]
        if cam_key is None:
            _check(lib().fbbev_lift_rank_build(*args, _stream()), 'fbbev_lift_rank_build')
        else:      # camera-keyed cache: device-side compare, early-out of the whole build when nothing changed
            if cam_key.numel() < cam_key_words(B, N) or cache_state.numel() < 2:
                raise FbbevError('cam_key / cache_state too small')
            _check(lib().fbbev_lift_rank_build_cached(*args, _dev(cam_key, I32, 'cam_key'), _dev(cache_state, I32, 'cache_state'),
                                                      _stream()), 'fbbev_lift_rank_build_cached')


def cam_key_words(B, N):
    return int(lib().fbbev_cam_key_words(int(B), int(N)))


def lift_splat_fused_ws_bytes(B, N, D, H, W, C, Z, Y, X):
    return int(lib().fbbev_lift_splat_fused_ws_bytes(B, N, D, H, W, C, Z, Y, X))


def lift_splat_fused_ws_views(workspace, B, N, D, H, W, C, Z, Y, X):
    """The index tensors fbbev_lift_splat_fused left in `workspace` (uint8), as int32 views of the padded arrays + counts + the NHWC
    feature rows: dict(ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths, interval_rank, counts, feat)."""
    off = (ctypes.c_size_t * 8)()
    _check(lib().fbbev_lift_splat_fused_ws_offsets(B, N, D, H, W, C, Z, Y, X, ctypes.cast(off, c_void_p)), 'fbbev_lift_splat_fused_ws_offsets')
    n = B * N * D * H * W
    names = ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths', 'interval_rank')
    out = {k: workspace[off[i]:off[i] + 4 * n].view(torch.int32) for i, k in enumerate(names)}
    out['counts'] = workspace[off[6]:off[6] + 8].view(torch.int32)
    out['feat'] = workspace[off[7]:off[7] + 4 * B * N * H * W * C].view(torch.float32).view(B, N, H, W, C)
    return out


def lift_splat_fused(xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, depth, context, lower3, interval3, grid_size3,
                     grid_zyx, out, workspace, tile_voxels=128, flags=None, frustum=None, cam_key=None, cache_state=None):
    """fbbev_lift_splat_fused: camera tensors + depth (B,N,D,H,W) + context (B,N,C,H,W) -> out (B,C,Z,Y,X) f32 / bf16 / f16 in ONE
    C call (geometry, ranking, NCHW->NHWC, tile index, dense pooling on the current stream).  workspace: uint8 tensor of
    lift_splat_fused_ws_bytes(...).  cam_key (int32[cam_key_words], -1) + cache_state (int32[4] = [0, 0, -1, -1]): camera-keyed cache."""
    B, N, D, H, W = depth.shape
    C = context.shape[2]
    Z, Y, X = grid_zyx
    if tuple(context.shape) != (B, N, C, H, W) or (ds.numel(), ys.numel(), xs.numel()) != (D, H, W):
        raise FbbevError('lift_splat_fused: depth (B,N,D,H,W), context (B,N,C,H,W), frustum axes (W), (H), (D)')
    if tuple(out.shape) != (B, C, Z, Y, X) or out.stride()[2:] != (Y * X, X, 1) or out.dtype not in (F32, torch.bfloat16, torch.float16):
        raise FbbevError('lift_splat_fused: out must be a (B,C,Z,Y,X) f32 / bf16 / f16 tensor with a contiguous (Z,Y,X) block')
    fl = (DEFAULT_POOL_FLAGS if flags is None else int(flags)) & ~(POOL_OUT_BF16 | POOL_OUT_F16 | POOL_CHANNELS_LAST)
    fl |= {F32: 0, torch.bfloat16: POOL_OUT_BF16, torch.float16: POOL_OUT_F16}[out.dtype]
    if (cam_key is None) != (cache_state is None):
        raise FbbevError('lift_splat_fused: cam_key and cache_state go together')
    if cam_key is not None and (cam_key.numel() < cam_key_words(B, N) or cache_state.numel() < 4):
        raise FbbevError('lift_splat_fused: cam_key / cache_state too small (cache_state: 4 x int32)')
    arr = ctypes.c_float * 3
    lo, it, gs = arr(*lower3), arr(*interval3), arr(*grid_size3)
    with _on(depth):
        _check(lib().fbbev_lift_splat_fused(
            _dev(frustum, F32, 'frustum') if frustum is not None else c_void_p(0), _dev(xs, F32, 'xs'), _dev(ys, F32, 'ys'),
            _dev(ds, F32, 'ds'), _dev(rots, F32, 'rots'), _dev(trans, F32, 'trans'), _dev(intrins, F32, 'intrins'),
            _dev(post_rots, F32, 'post_rots'), _dev(post_trans, F32, 'post_trans'), _dev(bda, F32, 'bda'), _dev(depth, F32, 'depth'),
            _dev(context, F32, 'context'), B, N, D, H, W, C, ctypes.cast(lo, c_void_p), ctypes.cast(it, c_void_p),
            ctypes.cast(gs, c_void_p), Z, Y, X, c_void_p(out.data_ptr()), out.stride(0), out.stride(1), int(tile_voxels), fl,
            c_void_p(workspace.data_ptr()), workspace.numel() * workspace.element_size(),
            _dev(cam_key, I32, 'cam_key') if cam_key is not None else c_void_p(0),
            _dev(cache_state, I32, 'cache_state') if cache_state is not None else c_void_p(0), _stream()), 'fbbev_lift_splat_fused')
    return out


def pool_dense_workspace_bytes(B, Z, Y, X):
    return int(lib().fbbev_pool_dense_workspace_bytes(B, Z, Y, X))


POOL_STORE_PLAIN, POOL_STORE_NT, POOL_STORE_SC1_NT, POOL_CPL8 = 0, 1, 4, 4


def pool_flags(store=4, cpl8=True, csplit=2, wg=256, swizzle=True, swz_log2=4):
    """Tuning flags of fbbev_bev_pool_v2_dense_fwd (include/fbbev.h); none changes the result bits."""
    cs = 0xF if csplit == 20 else (csplit & 0xF)
    return (store & 3) | (((store >> 2) & 1) << 17) | (POOL_CPL8 if cpl8 else 0) | (cs << 4) | ({256: 0, 128: 1}[wg] << 8) | \
        (0x400 if swizzle else 0) | ((swz_log2 & 0x1F) << 12)


# measured best PER LAUNCH on MI355X at BASELINE configs[1], B=16 (profiles/r01_pool_*.jsonl): 128-voxel
# tiles, channel range split over 2 workgroups, 8 channels/lane, `sc1 nt` stores (do not evict the
# gathered inputs from L2), chunks of 16 tiles dealt round-robin to the XCDs -> 0.56 ms = 0.745 of 8 TB/s
# (a linear torch.zero_ of the same buffer: 0.47 ms = 0.86).
DEFAULT_POOL_FLAGS = pool_flags()
DEFAULT_TILE_VOXELS = 128


POOL_CHANNELS_LAST = 0x100000
POOL_OUT_BF16, POOL_OUT_F16 = 0x800000, 0x1000000
POOL_PIPE = 0x4000000         # a workgroup walks a run of tiles with the next tile's staging loads in flight (dense grids; same bits)
POOL_GATHER8 = 0x8000000      # eight points per gather batch (experiment knob for dense grids; same bits)
POOL_SPLIT_LONG = 0x2000000   # tolerance mode: long intervals summed by the whole workgroup (<= 1e-4, not bit-exact)


def pool_tile_index(interval_rank, interval_starts, counts, n_intervals_max, B, Z, Y, X, tile_ws,
                    tile_voxels=64, flags=0, cache_state=None, table_gate=None):
    """cache_state + table_gate (int32[2], one per table, initialised to -1): camera-keyed cache, the table is kept only
    when the index set is unchanged AND this table was built for that build (fbbev_pool_tile_index_cached)."""
    if (cache_state is None) != (table_gate is None):
        raise FbbevError('pool_tile_index: cache_state and table_gate go together')
    with _on(interval_rank):
        args = [_dev(interval_rank, I32, 'interval_rank'), _dev(interval_starts, I32, 'interval_starts'),
                _dev(counts, I32, 'counts'), int(n_intervals_max), B, Z, Y, X, int(tile_voxels), int(flags),
                c_void_p(tile_ws.data_ptr()), tile_ws.numel() * tile_ws.element_size()]
        if cache_state is None:
            _check(lib().fbbev_pool_tile_index(*args, _stream()), 'fbbev_pool_tile_index')
        else:
            _check(lib().fbbev_pool_tile_index_cached(*args, _dev(cache_state, I32, 'cache_state'),
                                                      _dev(table_gate, I32, 'table_gate'), _stream()),
                   'fbbev_pool_tile_index_cached')


def pool_zmean(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
               out_mean, tile_ws, tile_voxels=64, flags=DEFAULT_POOL_FLAGS, z_groups=1, partial=None):
    """out_mean (B,C,Y,X) f32 contiguous = mean over z of the pooled sums (tile index of the same tile_voxels in tile_ws).
    z_groups > 1: the planes of a tile go to z_groups workgroups (fbbev_pool_zmean_split); partial = f32 buffer of >= z_groups * out numel."""
    if tuple(out_mean.shape) != (B, C, Y, X):
        raise FbbevError('out_mean must be (B,C,Y,X)')
    fl = int(flags) & ~(POOL_OUT_BF16 | POOL_OUT_F16 | POOL_SPLIT_LONG | POOL_PIPE)
    if z_groups > 1:
        if partial is None or partial.dtype != F32 or partial.numel() < z_groups * out_mean.numel():
            raise FbbevError('pool_zmean: z_groups > 1 needs a float32 partial buffer of z_groups * B*C*Y*X elements')
        with _on(depth):
            _check(lib().fbbev_pool_zmean_split(
                _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'), _dev(ranks_depth, I32, 'ranks_depth'),
                _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_rank, I32, 'interval_rank'),
                _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
                B, C, Z, Y, X, _dev(out_mean, F32, 'out_mean'), c_void_p(tile_ws.data_ptr()),
                tile_ws.numel() * tile_ws.element_size(), int(tile_voxels), fl, int(z_groups), c_void_p(partial.data_ptr()),
                partial.numel() * 4, _stream()), 'fbbev_pool_zmean_split')
        return out_mean
    with _on(depth):
        _check(lib().fbbev_pool_zmean(
            _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'), _dev(ranks_depth, I32, 'ranks_depth'),
            _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_rank, I32, 'interval_rank'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
            B, C, Z, Y, X, _dev(out_mean, F32, 'out_mean'), c_void_p(tile_ws.data_ptr()),
            tile_ws.numel() * tile_ws.element_size(), int(tile_voxels), int(flags) & ~(POOL_OUT_BF16 | POOL_OUT_F16 | POOL_SPLIT_LONG | POOL_PIPE),
            _stream()), 'fbbev_pool_zmean')
    return out_mean


def pool_zmean_rows(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
                    out_rows, tile_ws, tile_voxels=64, flags=DEFAULT_POOL_FLAGS, row_bias=None):
    """out_rows (B, Y*X, C) f32 contiguous = mean over z of the pooled sums + row_bias (Y*X, C): the Z-mean written as the backward
    projection's query rows (fbbev_pool_zmean_rows; single pass)."""
    if tuple(out_rows.shape) != (B, Y * X, C) or not out_rows.is_contiguous():
        raise FbbevError('out_rows must be (B, Y*X, C) contiguous')
    if row_bias is not None and (tuple(row_bias.shape) != (Y * X, C) or not row_bias.is_contiguous()):
        raise FbbevError('row_bias must be (Y*X, C) contiguous')
    with _on(depth):
        _check(lib().fbbev_pool_zmean_rows(
            _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'), _dev(ranks_depth, I32, 'ranks_depth'),
            _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_rank, I32, 'interval_rank'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
            B, C, Z, Y, X, _dev(row_bias, F32, 'row_bias') if row_bias is not None else None, _dev(out_rows, F32, 'out_rows'),
            c_void_p(tile_ws.data_ptr()), tile_ws.numel() * tile_ws.element_size(), int(tile_voxels),
            int(flags) & ~(POOL_OUT_BF16 | POOL_OUT_F16 | POOL_SPLIT_LONG | POOL_PIPE), _stream()), 'fbbev_pool_zmean_rows')
    return out_rows


def bev_pool_v2_dense_fwd(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts,
                          interval_lengths, B, C, Z, Y, X, out, tile_ws, tile_voxels=64,
                          flags=DEFAULT_POOL_FLAGS, addend=None):
    """`out` is (B,C,Z,Y,X) f32 whose (Z,Y,X) block is contiguous (batch/channel strides may be padded), or,
    with POOL_CHANNELS_LAST in `flags`, a contiguous (B,Z,Y,X,C) tensor (the reference op's own layout)."""
    if not out.is_cuda or out.dtype not in (F32, torch.bfloat16, torch.float16):
        raise FbbevError('out must be a GPU float32 / bfloat16 / float16 tensor')
    flags = int(flags) & ~(POOL_OUT_BF16 | POOL_OUT_F16)
    flags |= {F32: 0, torch.bfloat16: POOL_OUT_BF16, torch.float16: POOL_OUT_F16}[out.dtype]   # storage type follows `out`
    if flags & POOL_CHANNELS_LAST:
        if tuple(out.shape) != (B, Z, Y, X, C) or not out.is_contiguous():
            raise FbbevError('channels-last out must be a contiguous (B,Z,Y,X,C) tensor')
        sb, sc = C * Z * Y * X, Z * Y * X
    else:
        if tuple(out.shape) != (B, C, Z, Y, X) or out.stride()[2:] != (Y * X, X, 1):
            raise FbbevError('out must be a (B,C,Z,Y,X) tensor with a contiguous (Z,Y,X) block')
        sb, sc = out.stride(0), out.stride(1)
    args = (_dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'), _dev(ranks_depth, I32, 'ranks_depth'),
            _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_rank, I32, 'interval_rank'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
            B, C, Z, Y, X, c_void_p(out.data_ptr()), sb, sc, c_void_p(tile_ws.data_ptr()),
            tile_ws.numel() * tile_ws.element_size(), int(tile_voxels), int(flags))
    with _on(depth):
        if addend is None:
            _check(lib().fbbev_bev_pool_v2_dense_fwd(*args, _stream()), 'fbbev_bev_pool_v2_dense_fwd')
        else:       # out = pooled + addend[b,c,y,x] broadcast over z (the re-add of fbocc.py:365-366)
            if tuple(addend.shape) != (B, C, Y, X):
                raise FbbevError('addend must be (B,C,Y,X)')
            _check(lib().fbbev_bev_pool_v2_dense_fwd_add(*args, _dev(addend, F32, 'addend'), _stream()),
                   'fbbev_bev_pool_v2_dense_fwd_add')


def diag_pool_store_floor(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
                          out, tile_ws, tile_voxels, flags, mode):
    """Measurement aid (never on the product path): the default fp32 (or, for a bfloat16 `out`, bf16-storage) dense-kernel
    instantiation with parts compiled out -- mode 1 the store pattern alone, mode 2 all but the depth / feature gathers, mode 3 all
    but the stores.  `out` receives zeros (mode 3: untouched)."""
    if tuple(out.shape) != (B, C, Z, Y, X) or not out.is_contiguous() or out.dtype not in (F32, torch.bfloat16):
        raise FbbevError('out must be a contiguous (B,C,Z,Y,X) float32 / bfloat16 tensor')
    flags = (int(flags) & ~(POOL_OUT_BF16 | POOL_OUT_F16)) | (POOL_OUT_BF16 if out.dtype == torch.bfloat16 else 0)
    with _on(depth):
        _check(lib().fbbev_diag_pool_store_floor(
            _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'), _dev(ranks_depth, I32, 'ranks_depth'),
            _dev(ranks_feat, I32, 'ranks_feat'), _dev(interval_rank, I32, 'interval_rank'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(interval_lengths, I32, 'interval_lengths'),
            B, C, Z, Y, X, c_void_p(out.data_ptr()), c_void_p(tile_ws.data_ptr()), tile_ws.numel() * tile_ws.element_size(),
            int(tile_voxels), int(flags), int(mode), _stream()), 'fbbev_diag_pool_store_floor')


def pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X):
    return int(lib().fbbev_pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X))


def bev_pool_v2_dense_bwd(out_grad, depth, feat, ranks_depth, interval_rank, interval_starts, counts,
                          n_intervals_max, grid_zyx, depth_grad, feat_grad, workspace, zgrad=None, zscale=0.0):
    """Sync-free backward of the fused lift-splat.  out_grad: (B,C,Z,Y,X) f32 with a contiguous (Z,Y,X)
    block; depth (B,N,D,H,W); feat (B,N,H,W,C); depth_grad / feat_grad: same shapes, written completely.
    zgrad (B,C,Y,X), optional: added to every z plane of out_grad scaled by zscale (the Z-mean's backward, folded in)."""
    B, N, D, H, W = depth.shape
    C = feat.shape[-1]
    Z, Y, X = grid_zyx
    if tuple(out_grad.shape) != (B, C, Z, Y, X) or out_grad.stride()[2:] != (Y * X, X, 1):
        raise FbbevError('out_grad must be a (B,C,Z,Y,X) tensor with a contiguous (Z,Y,X) block')
    if tuple(feat.shape) != (B, N, H, W, C) or depth_grad.shape != depth.shape or feat_grad.shape != feat.shape:
        raise FbbevError('depth/feat/grad shapes do not match')
    if zgrad is not None:
        if tuple(zgrad.shape) != (B, C, Y, X):
            raise FbbevError('zgrad must be (B,C,Y,X)')
        with _on(depth):
            _check(lib().fbbev_bev_pool_v2_dense_bwd_z(
                _dev(out_grad, F32, 'out_grad', contiguous=False), out_grad.stride(0), out_grad.stride(1),
                _dev(zgrad, F32, 'zgrad'), float(zscale), _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'),
                _dev(ranks_depth, I32, 'ranks_depth'), _dev(interval_rank, I32, 'interval_rank'),
                _dev(interval_starts, I32, 'interval_starts'), _dev(counts, I32, 'counts'), int(n_intervals_max),
                B, N, D, H, W, C, Z, Y, X, _dev(depth_grad, F32, 'depth_grad'), _dev(feat_grad, F32, 'feat_grad'),
                c_void_p(workspace.data_ptr()), workspace.numel() * workspace.element_size(), _stream()),
                'fbbev_bev_pool_v2_dense_bwd_z')
        return
    with _on(depth):
        _check(lib().fbbev_bev_pool_v2_dense_bwd(
            _dev(out_grad, F32, 'out_grad', contiguous=False), out_grad.stride(0), out_grad.stride(1),
            _dev(depth, F32, 'depth'), _dev(feat, F32, 'feat'),
            _dev(ranks_depth, I32, 'ranks_depth'), _dev(interval_rank, I32, 'interval_rank'),
            _dev(interval_starts, I32, 'interval_starts'), _dev(counts, I32, 'counts'), int(n_intervals_max),
            B, N, D, H, W, C, Z, Y, X, _dev(depth_grad, F32, 'depth_grad'), _dev(feat_grad, F32, 'feat_grad'),
            c_void_p(workspace.data_ptr()), workspace.numel() * workspace.element_size(), _stream()),
            'fbbev_bev_pool_v2_dense_bwd')


def msda_fwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out):
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = sampling_loc.shape
    with _on(value):
        _check(lib().fbbev_msda_fwd(
            _dev(value, F32, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
            _dev(level_start_index, I64, 'level_start_index'), _dev(sampling_loc, F32, 'sampling_loc'),
            _dev(attn_weight, F32, 'attn_weight'), B, S, M, Dh, L, Q, P, _dev(out, F32, 'out'),
            _stream()), 'fbbev_msda_fwd')


def msda_fwd_fused(value, spatial_shapes, level_start_index, ref_points, offsets, attn_weight, out, head_dim=None,
                   offsets_head_minor=False, value_interleaved=False):
    """value (B,S,M,HS); ref_points (B,Q,L,2); offsets (B,Q,M,L,P,2) or head-minor (B,Q,L,P,M,2); attn (B,Q,M,L,P).
    value_interleaved: a token's M*HS floats are stored (HS/4, M, 4) -- chunk-major -- instead of (M, HS)."""
    B, S, M, HS = value.shape
    Dh = HS if head_dim is None else int(head_dim)
    _, Q, _, L, P = attn_weight.shape
    with _on(value):
        _check(lib().fbbev_msda_fwd_fused(
            _dev(value, F32, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
            _dev(level_start_index, I64, 'level_start_index'), _dev(ref_points, F32, 'ref_points'),
            _dev(offsets, F32, 'offsets'), _dev(attn_weight, F32, 'attn_weight'), B, S, M, Dh, L, Q, P, HS,
            (1 if offsets_head_minor else 0) | (4 if value_interleaved else 0), _dev(out, F32, 'out'), _stream()),
            'fbbev_msda_fwd_fused')
    return out


def msda_bwd_ws_bytes(B, S, M, Dh, L, Q, P, level_hw=None):
    """Scratch bytes of the atomic-free backward (0: not available for this shape / no host level shapes)."""
    if level_hw is None:
        return 0
    return lib().fbbev_msda_bwd_ws_bytes(B, S, M, Dh, L, Q, P, _level_hw(level_hw, L))


def msda_bwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
             grad_value, grad_sampling_loc, grad_attn_weight, level_hw=None):
    """grad_sampling_loc / grad_attn_weight pre-zeroed (accumulated into).  With `level_hw` (host (h, w) per level) and a
    shape the band-binned kernels take, grad_value is WRITTEN (fixed-point LDS planes, bit-reproducible, no pre-zeroing
    needed); otherwise it is accumulated into with fp32 global atomics and must be pre-zeroed -- callers that pass
    level_hw check msda_bwd_ws_bytes() to know which."""
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = sampling_loc.shape
    with _on(value):
        args = (_dev(value, F32, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
                _dev(level_start_index, I64, 'level_start_index'), _dev(sampling_loc, F32, 'sampling_loc'),
                _dev(attn_weight, F32, 'attn_weight'), _dev(grad_output, F32, 'grad_output'),
                B, S, M, Dh, L, Q, P, _dev(grad_value, F32, 'grad_value'),
                _dev(grad_sampling_loc, F32, 'grad_sampling_loc'), _dev(grad_attn_weight, F32, 'grad_attn_weight'))
        need = msda_bwd_ws_bytes(B, S, M, Dh, L, Q, P, level_hw)
        if need:
            ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=value.device)
            _check(lib().fbbev_msda_bwd_ws(*args, _level_hw(level_hw, L), ws.data_ptr(), need, _stream()), 'fbbev_msda_bwd_ws')
        else:
            _check(lib().fbbev_msda_bwd(*args, _stream()), 'fbbev_msda_bwd')


def da_cross_attn_fwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                      attn, d0, dstep, slots, head_minor=0, head_dim=None, zero_token=False, bev_w=0):
    """value (B*Ncam,S,M,Dh); pred_depth (B*Ncam,DC,H0,W0); ref_cam (Ncam,B,Q,Za,2); mask (Ncam,B,Q,Za) bool;
    qdepth (Ncam,B,Q,Za); offsets (B,Q,M,L,P,2); attn (B,Q,M,L,P); head_minor bit 0: offsets is (B,Q,L,P,M,2),
    bit 1: attn is (B,Q,L,P,M), bit 2: a value token's M*HS floats are stored (HS/4, M, 4); slots (B,Q,M*Dh).
    zero_token=True (fp32 rows): `value` is the first B*Ncam*S tokens of a buffer that holds one more, all-zero token right
    behind them (da_value_buffer) -- the pipelined sampler fbbev_da_cross_attn_fwd_zt."""
    Ncam, B, Q, Za = mask.shape
    _, S, M, HS = value.shape                 # HS = head stride; head_dim (<= HS) of them are channels, the rest padding
    Dh = HS if head_dim is None else int(head_dim)
    head_minor = int(head_minor)
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    if attn.shape[-1 if head_minor & 2 else 2] != M or offsets.shape[-2 if head_minor & 1 else 2] != M:
        raise FbbevError('offsets / attn layout does not match head_minor')
    if tuple(slots.shape[-1:]) != (M * Dh,):
        raise FbbevError('slots must be (B,Q,M*head_dim)')
    DC = pred_depth.shape[1]
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if value.dtype != torch.float32:
        with _on(value):
            _check(lib().fbbev_da_cross_attn_fwd_e(
                _dev(value, value.dtype, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
                _dev(level_start_index, I64, 'level_start_index'), _dev(pred_depth, F32, 'pred_depth'),
                _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'), _dev(qdepth, F32, 'qdepth'),
                _dev(offsets, F32, 'offsets'), _dev(attn, F32, 'attn'), B, Ncam, S, M, Dh, L, Q, P, Za, DC,
                float(d0), float(dstep), head_minor, HS, ELEM_TYPE[value.dtype], _dev(slots, F32, 'slots'), _stream()),
                'fbbev_da_cross_attn_fwd_e')
        return
    fn, name, extra = lib().fbbev_da_cross_attn_fwd, 'fbbev_da_cross_attn_fwd', ()
    if zero_token:
        extra = (int(bev_w) if bev_w and Q % int(bev_w) == 0 else 0,)
        need = (value.storage_offset() + value.numel() + M * HS) * 4
        if not value.is_contiguous() or value.untyped_storage().nbytes() < need:
            raise FbbevError('zero_token=True needs the value view of da_value_buffer (one extra token behind the rows)')
        fn, name = lib().fbbev_da_cross_attn_fwd_zt, 'fbbev_da_cross_attn_fwd_zt'
    with _on(value):
        _check(fn(
            _dev(value, F32, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
            _dev(level_start_index, I64, 'level_start_index'), _dev(pred_depth, F32, 'pred_depth'),
            _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'), _dev(qdepth, F32, 'qdepth'),
            _dev(offsets, F32, 'offsets'), _dev(attn, F32, 'attn'), B, Ncam, S, M, Dh, L, Q, P, Za, DC,
            float(d0), float(dstep), head_minor, HS, *extra, _dev(slots, F32, 'slots'), _stream()), name)


DA_ATTN_LOGITS = 0x10          # head_minor flag of the zero-token entry: `attn` holds raw logits, softmax fused in the kernel


def da_fuses_softmax(B, Ncam, S, M, Dh, L, Q, P, Za, head_minor, HS):
    """True when fbbev_da_cross_attn_fwd_zt takes the pipelined kernel for this shape, i.e. may be handed raw
    attention logits (head_minor | DA_ATTN_LOGITS)."""
    return bool(lib().fbbev_da_cross_attn_fwd_zt_fuses_softmax(B, Ncam, S, M, Dh, L, Q, P, Za, int(head_minor), HS))


def da_value_buffer(tokens, row_floats, device):
    """(buffer, rows): `rows` = the (tokens, row_floats) fp32 view the value projection writes into, `buffer` holds one more
    token behind it, zeroed here (the zero token of fbbev_da_cross_attn_fwd_zt)."""
    buf = torch.empty((tokens + 1, row_floats), dtype=F32, device=device)
    buf[tokens].zero_()
    return buf, buf[:tokens]


def _level_hw(level_hw, L):
    """host (h, w) pairs -> ctypes int32 array (or None): lets the backward plan LDS token regions without a device read"""
    if level_hw is None:
        return None
    flat = [int(x) for hw in level_hw for x in hw]
    if len(flat) != 2 * L:
        raise FbbevError('level_hw must hold num_levels (h, w) pairs')
    return (c_int32 * len(flat))(*flat)


def da_cross_attn_bwd_ws_bytes(B, Ncam, S, M, Dh, Q, HS, L, P, level_hw=None, Za=None):
    """Za given: the size of the route a launch with that many Z anchors takes; None: the maximum over both LDS-plane routes."""
    arr = _level_hw(level_hw, L)
    if Za is not None:
        return lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B, Ncam, S, M, Dh, Q, HS, L, P, int(Za), arr)
    return lib().fbbev_da_cross_attn_bwd_ws_bytes(B, Ncam, S, M, Dh, Q, HS, L, P, arr)


def da_cross_attn_bwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn,
                      grad_slots, d0, dstep, head_minor, grad_value, grad_pred_depth, grad_offsets, grad_attn, head_dim=None,
                      lds_planes=True, level_hw=None, bev_w=0):
    """Backward of da_cross_attn_fwd; the four grad tensors must be pre-zeroed (accumulated into).  bev_w: width of the BEV grid
    the Q queries form (0 = a plain list): lets the unit-gradient kernel take 8 x 8 patches of it."""
    Ncam, B, Q, Za = mask.shape
    _, S, M, HS = value.shape
    Dh = HS if head_dim is None else int(head_dim)
    head_minor = int(head_minor)
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    DC = pred_depth.shape[1]
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    args = (_dev(value, F32, 'value'), _dev(spatial_shapes, I64, 'spatial_shapes'),
            _dev(level_start_index, I64, 'level_start_index'), _dev(pred_depth, F32, 'pred_depth'),
            _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'), _dev(qdepth, F32, 'qdepth'),
            _dev(offsets, F32, 'offsets'), _dev(attn, F32, 'attn'), _dev(grad_slots, F32, 'grad_slots'),
            B, Ncam, S, M, Dh, L, Q, P, Za, DC, float(d0), float(dstep), head_minor, HS,
            _dev(grad_value, F32, 'grad_value'), _dev(grad_pred_depth, F32, 'grad_pred_depth'),
            _dev(grad_offsets, F32, 'grad_offsets'), _dev(grad_attn, F32, 'grad_attn'))
    with _on(value):
        # value gradient through fixed-point LDS planes when the shape fits (fbbev_da_cross_attn_bwd_ws: output-owned planes + hit
        # lists, or query chunks + partial planes for small launches), else the global-atomic kernel
        arr = _level_hw(level_hw, L)
        need = lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B, Ncam, S, M, Dh, Q, HS, L, P, Za, arr) if lds_planes else 0
        if need:
            if any(t.data_ptr() % 8 for t in (offsets, grad_offsets, grad_slots)):       # the owned route's alignment: size for both
                need = max(need, lib().fbbev_da_cross_attn_bwd_ws_bytes(B, Ncam, S, M, Dh, Q, HS, L, P, arr))
            ws = torch.empty(need // 4, dtype=torch.float32, device=value.device)
            _check(lib().fbbev_da_cross_attn_bwd_ws_grid(*args, arr, ws.data_ptr(), need, int(bev_w or 0), _stream()),
                   'fbbev_da_cross_attn_bwd_ws_grid')
        else:
            _check(lib().fbbev_da_cross_attn_bwd(*args, _stream()), 'fbbev_da_cross_attn_bwd')


def da_cross_attn_bwd_planes_supported(B, Ncam, S, M, Dh, L, Q, P, Za, HS, level_hw, bev_w):
    if level_hw is None or not bev_w:
        return False
    return bool(lib().fbbev_da_cross_attn_bwd_planes_supported(B, Ncam, S, M, Dh, L, Q, P, Za, HS, _level_hw(level_hw, L), int(bev_w)))


def da_cross_attn_bwd_planes(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots,
                             d0, dstep, head_minor, HS, grad_value, grad_pred_depth, grad_offsets, grad_attn, level_hw, bev_w):
    """fbbev_da_cross_attn_bwd_planes: the DA backward on the forward's head planes; grad_value (B*Ncam, S, M, HS), grad_offsets and
    grad_attn are written in full (torch.empty is enough), grad_pred_depth must be zeroed."""
    Ncam, B, Q, Za = mask.shape
    BN, M, S, Dh = planes.shape
    head_minor = int(head_minor)
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    DC = pred_depth.shape[1]
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    arr = _level_hw(level_hw, L)
    need = lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B, Ncam, S, M, Dh, Q, HS, L, P, Za, arr)
    ws = torch.empty(max(need, 16) // 4, dtype=F32, device=planes.device)
    with _on(planes):
        _check(lib().fbbev_da_cross_attn_bwd_planes(
            _dev(planes, F32, 'planes'), _dev(spatial_shapes, I64, 'spatial_shapes'), _dev(level_start_index, I64, 'level_start_index'),
            _dev(pred_depth, F32, 'pred_depth'), _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'), _dev(qdepth, F32, 'qdepth'),
            _dev(offsets, F32, 'offsets'), _dev(attn, F32, 'attn'), _dev(grad_slots, F32, 'grad_slots'), B, Ncam, S, M, Dh, L, Q, P, Za,
            DC, float(d0), float(dstep), head_minor, int(HS), _dev(grad_value, F32, 'grad_value'),
            _dev(grad_pred_depth, F32, 'grad_pred_depth'), _dev(grad_offsets, F32, 'grad_offsets'), _dev(grad_attn, F32, 'grad_attn'),
            arr, ws.data_ptr(), need, int(bev_w), _stream()), 'fbbev_da_cross_attn_bwd_planes')


def point_sampling(xs, ys, zs, rots, trans, intrins, post_rots, post_trans, bda, ogfH, ogfW, ref_cam, mask, qdepth):
    """ref_cam (N,B,Y*X,Za,2) f32, mask (N,B,Y*X,Za) uint8/bool, qdepth (N,B,Y*X,Za) f32, all preallocated."""
    B, N = trans.shape[:2]
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    with _on(ref_cam):
        _check(lib().fbbev_point_sampling(
            _dev(xs, F32, 'xs'), _dev(ys, F32, 'ys'), _dev(zs, F32, 'zs'), _dev(rots, F32, 'rots'),
            _dev(trans, F32, 'trans'), _dev(intrins, F32, 'intrins'), _dev(post_rots, F32, 'post_rots'),
            _dev(post_trans, F32, 'post_trans'), _dev(bda, F32, 'bda'), B, N, ys.numel(), xs.numel(), zs.numel(),
            float(ogfH), float(ogfW), _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'),
            _dev(qdepth, F32, 'qdepth'), _stream()), 'fbbev_point_sampling')


def nchw_to_nhwc(context):
    """context (B,N,C,H,W) f32 contiguous -> feat (B,N,H,W,C) contiguous (LDS-tiled transpose kernel)."""
    B, N, C, H, W = context.shape
    feat = torch.empty((B, N, H, W, C), dtype=torch.float32, device=context.device)
    with _on(context):
        _check(lib().fbbev_nchw_to_nhwc(_dev(context, F32, 'context'), _dev(feat, F32, 'feat'), B * N, C, H * W,
                                        _stream()), 'fbbev_nchw_to_nhwc')
    return feat


def tokens_from_nchw(x, out, out_offset=0, bias=None, pos_bias=None):
    """x (n_images, C, HW) f32 contiguous -> out[img, out_offset/C + p, c] = x[img, c, p] (+ bias[img % rows, c]) or, with
    pos_bias (HW, C), + pos_bias[p, c]; out (n_images, S, C) contiguous with S >= HW rows per image; out_offset in floats."""
    n, C, HW = x.shape
    if out.shape[0] != n or out.shape[-1] != C or not out.is_contiguous():
        raise FbbevError('tokens_from_nchw: out must be (n_images, S, C) contiguous')
    stride = out.stride(0)
    if out_offset + C * HW > stride:
        raise FbbevError('tokens_from_nchw: level does not fit the token rows')
    if pos_bias is not None:
        if bias is not None or tuple(pos_bias.shape) != (HW, C):
            raise FbbevError('tokens_from_nchw: pos_bias is (HW, C) and excludes bias')
        with _on(x):
            _check(lib().fbbev_tokens_from_nchw_pos(_dev(x, F32, 'x'), _dev(out, F32, 'out'), n, C, HW, stride, int(out_offset),
                                                    _dev(pos_bias, F32, 'pos_bias'), _stream()), 'fbbev_tokens_from_nchw_pos')
        return out
    with _on(x):
        _check(lib().fbbev_tokens_from_nchw(
            _dev(x, F32, 'x'), _dev(out, F32, 'out'), n, C, HW, stride, int(out_offset),
            None if bias is None else _dev(bias, F32, 'bias'), 0 if bias is None else bias.shape[0], _stream()),
            'fbbev_tokens_from_nchw')
    return out


def tokens_from_nchw_levels(levels, out, bias=None):
    """levels: list of (n_images, C, HW_l) f32 contiguous tensors (<= 8) -> out (n_images, sum HW_l, C) contiguous, level l at rows
    [start_l, start_l + HW_l); + bias[img % rows, c] when given.  ONE launch for the whole pyramid (fbbev_tokens_from_nchw_levels)."""
    n, C = levels[0].shape[:2]
    hws = [int(t.shape[2]) for t in levels]
    if any(t.shape[0] != n or t.shape[1] != C for t in levels) or tuple(out.shape) != (n, sum(hws), C) or not out.is_contiguous():
        raise FbbevError('tokens_from_nchw_levels: levels are (n_images, C, HW_l); out is (n_images, sum HW_l, C) contiguous')
    ptrs = (c_void_p * len(levels))(*[_dev(t, F32, 'level').value for t in levels])
    hw = (c_int32 * len(levels))(*hws)
    with _on(out):
        _check(lib().fbbev_tokens_from_nchw_levels(ptrs, hw, len(levels), _dev(out, F32, 'out'), n, C,
                                                   None if bias is None else _dev(bias, F32, 'bias'),
                                                   0 if bias is None else bias.shape[0], _stream()), 'fbbev_tokens_from_nchw_levels')
    return out


def transpose_last2(x):
    """(B, R, S) f32 contiguous -> (B, S, R) contiguous, LDS-tiled (torch's .transpose().contiguous() runs its
    generic strided copy at ~0.7 TB/s on these shapes)."""
    B, R, S = x.shape
    out = torch.empty((B, S, R), dtype=torch.float32, device=x.device)
    return tokens_from_nchw(x, out)


def history_flow(history_forward_augs, curr_to_prev_ego_rt, bda, dx3, lower3):
    """(B,4,4), (B,4,4), (B,3,3) f32 GPU tensors + host (x,y,z) voxel size / grid lower bound -> rt_flow (B,4,4)."""
    B = bda.shape[0]
    flow = torch.empty((B, 4, 4), dtype=torch.float32, device=bda.device)
    arr = ctypes.c_float * 3
    d, lo = arr(*[float(v) for v in dx3]), arr(*[float(v) for v in lower3])
    with _on(bda):
        _check(lib().fbbev_history_flow(_dev(history_forward_augs, F32, 'history_forward_augs'),
                                        _dev(curr_to_prev_ego_rt, F32, 'curr_to_prev_ego_rt'), _dev(bda, F32, 'bda'),
                                        ctypes.cast(d, c_void_p), ctypes.cast(lo, c_void_p), B, _dev(flow, F32, 'flow'),
                                        _stream()), 'fbbev_history_flow')
    return flow


ELEM_TYPE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}       # FBBEV_ELEM_*: storage element of the history ring


def history_warp(history, rt_flow, out):
    """history, out: (B,CH,Z,Y,X) f32 / bf16 / f16 (the same type) whose per-sample (CH,Z,Y,X) block is contiguous (batch
    stride free).  16-bit storage: fp32 taps math, rounded once at the store."""
    B, CH, Z, Y, X = history.shape
    if tuple(out.shape) != (B, CH, Z, Y, X):
        raise FbbevError('out must have the shape of history')
    if history.dtype not in ELEM_TYPE or out.dtype != history.dtype:
        raise FbbevError('history / out must both be f32, bf16 or f16')
    for t, n in ((history, 'history'), (out, 'out')):
        if t.stride()[1:] != (Z * Y * X, Y * X, X, 1):
            raise FbbevError(f'{n}: the (CH,Z,Y,X) block of a sample must be contiguous')
    with _on(history):
        _check(lib().fbbev_history_warp_e(_dev(history, history.dtype, 'history', contiguous=False), history.stride(0),
                                          _dev(rt_flow, F32, 'rt_flow'), B, CH, Z, Y, X,
                                          _dev(out, out.dtype, 'out', contiguous=False), out.stride(0),
                                          ELEM_TYPE[history.dtype], _stream()), 'fbbev_history_warp_e')
    return out


def history_warp_vm(history, rt_flow, out, grid_zyx):
    """Voxel-major ring: history, out (B,T,N,C) f32 / bf16 / f16 (the same type), N = Z*Y*X with x fastest, the (T,N,C) block
    of a sample contiguous (batch stride free).  Same taps, weights and roundings as history_warp: the same element bits."""
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    if tuple(out.shape) != (B, T, N, C) or N != Z * Y * X:
        raise FbbevError('history_warp_vm: out must have the shape of history, N = Z*Y*X')
    if history.dtype not in ELEM_TYPE or out.dtype != history.dtype:
        raise FbbevError('history / out must both be f32, bf16 or f16')
    for t, n in ((history, 'history'), (out, 'out')):
        if t.stride()[1:] != (N * C, C, 1):
            raise FbbevError(f'{n}: the (T,N,C) block of a sample must be contiguous')
    with _on(history):
        _check(lib().fbbev_history_warp_vm(_dev(history, history.dtype, 'history', contiguous=False), history.stride(0),
                                           _dev(rt_flow, F32, 'rt_flow'), B, T, C, Z, Y, X,
                                           _dev(out, out.dtype, 'out', contiguous=False), out.stride(0),
                                           ELEM_TYPE[history.dtype], _stream()), 'fbbev_history_warp_vm')
    return out


def history_fused_vm(history, rt_flow, nxt, grid_zyx, w1, bias1, w2, bias2, out):
    """Warp + new ring + both convolutions in one launch (fbbev_history_fused_vm): history (B,T,N,C) and nxt (B,T+1,N,C)
    voxel-major bf16 / f16 rings (nxt[:, 0] = the current frame, already stored), w1 (C,C), bias1 (B*(T+1), C), w2 (Cout,(T+1)C),
    bias2 (Cout), out (B,Cout,N) f32.  Writes nxt[:, 1:] (== history_warp_vm) and out (== history_conv(nxt, ..., bfloat16))."""
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    Cout = w2.shape[0]
    if tuple(nxt.shape) != (B, T + 1, N, C) or N != Z * Y * X or tuple(out.shape) != (B, Cout, N):
        raise FbbevError('history_fused_vm: nxt must be (B,T+1,N,C), out (B,Cout,N), N = Z*Y*X')
    if history.dtype not in (torch.bfloat16, torch.float16) or nxt.dtype != history.dtype:
        raise FbbevError('history_fused_vm: 16-bit rings of one type')
    for t, n in ((history, 'history'), (nxt, 'nxt')):
        if t.stride()[1:] != (N * C, C, 1):
            raise FbbevError(f'{n}: the (T,N,C) block of a sample must be contiguous')
    ws = torch.empty((2 + T) * C * max(C, Cout, 96), dtype=torch.float32, device=history.device)   # fragment-ordered weights
    with _on(history):
        _check(lib().fbbev_history_fused_vm(
            _dev(history, history.dtype, 'history', contiguous=False), history.stride(0),
            _dev(nxt, nxt.dtype, 'nxt', contiguous=False), nxt.stride(0), _dev(rt_flow, F32, 'rt_flow'), _dev(w1, F32, 'w1'),
            _dev(bias1, F32, 'bias1'), _dev(w2, F32, 'w2'), _dev(bias2, F32, 'bias2'), B, T, C, Cout, Z, Y, X,
            _dev(out, F32, 'out'), c_void_p(ws.data_ptr()), ws.numel() * 4, ELEM_TYPE[history.dtype], _stream()),
            'fbbev_history_fused_vm')
    return out


def history_frame_vm(curr, out, inner=1):
    """curr (B,C,N) f32 contiguous planes -> out (B,N,C) f32 / bf16 / f16 rows (batch stride free): one frame slot of a
    voxel-major ring, rounded once for 16-bit storage.  inner = Z: the planes are (Y,X,Z) volumes, the rows (Z,Y,X)-ordered."""
    B, C, N = curr.shape
    if tuple(out.shape) != (B, N, C) or out.stride()[1:] != (C, 1) or out.dtype not in ELEM_TYPE:
        raise FbbevError('history_frame_vm: out must be (B,N,C) rows')
    with _on(curr):
        _check(lib().fbbev_history_frame_vm(_dev(curr, F32, 'curr'), B, C, N, int(inner), _dev(out, out.dtype, 'out', contiguous=False),
                                            out.stride(0), ELEM_TYPE[out.dtype], _stream()), 'fbbev_history_frame_vm')
    return out


def layernorm(x, weight, bias, eps, residual=None, out=None):
    """LayerNorm over the last dim of a contiguous f32 GPU tensor (C % 4 == 0, C <= 128); out = LN(x [+ residual])."""
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    with _on(x):
        _check(lib().fbbev_layernorm(_dev(x, F32, 'x'), _dev(residual, F32, 'residual') if residual is not None else None,
                                     _dev(weight, F32, 'weight'), _dev(bias, F32, 'bias'), float(eps), rows, C,
                                     _dev(out, F32, 'out'), _stream()), 'fbbev_layernorm')
    return out


def rows_linear_x3_fragments(weight):
    """W (O, I) f32 -> the split bf16 MFMA fragments of fbbev_rows_linear_x3 (a uint8 buffer; build once per weight version)."""
    O, I = weight.shape
    need = lib().fbbev_rows_linear_x3_fragment_bytes(I, O)
    frag = torch.empty(need, dtype=torch.uint8, device=weight.device)
    with _on(weight):
        _check(lib().fbbev_rows_linear_x3_fragments(_dev(weight, F32, 'weight'), I, O, frag.data_ptr(), need, _stream()),
               'fbbev_rows_linear_x3_fragments')
    return frag


def rows_linear_x3(x, fragments, bias, out_features, relu=False, out=None, addend=None):
    """x (R, I) f32 rows (row stride >= I, unit column stride) -> out (R, O) = x W^T + bias (+ ReLU), split-operand bf16 MFMA.
    addend (P, I): the rows are x[r] + addend[r % P] (R % P == 0)."""
    R, I = x.shape
    if x.stride(1) != 1:
        raise FbbevError('rows_linear_x3: rows must have unit column stride')
    if out is None:
        out = torch.empty((R, out_features), dtype=F32, device=x.device)
    if out.shape != (R, out_features) or out.stride(1) != 1:
        raise FbbevError('rows_linear_x3: out must be (rows, out_features) with unit column stride')
    b = _dev(bias, F32, 'bias') if bias is not None else None
    with _on(x):
        if addend is None:
            _check(lib().fbbev_rows_linear_x3(_dev(x, F32, 'x', contiguous=False), x.stride(0), fragments.data_ptr(), b, R, I,
                                              out_features, 1 if relu else 0, _dev(out, F32, 'out', contiguous=False),
                                              out.stride(0), _stream()), 'fbbev_rows_linear_x3')
        else:
            if addend.dim() != 2 or addend.shape[1] != I or addend.stride(1) != 1 or R % addend.shape[0] != 0:
                raise FbbevError('rows_linear_x3: addend must be (P, in_features) rows with rows % P == 0')
            _check(lib().fbbev_rows_linear_x3_add(_dev(x, F32, 'x', contiguous=False), x.stride(0),
                                                  _dev(addend, F32, 'addend', contiguous=False), addend.stride(0), addend.shape[0],
                                                  fragments.data_ptr(), b, R, I, out_features, 1 if relu else 0,
                                                  _dev(out, F32, 'out', contiguous=False), out.stride(0), _stream()),
                   'fbbev_rows_linear_x3_add')
    return out


def rows_linear_x3_ln(x, fragments, bias, out_features, residual, ln_weight, ln_bias, eps, out=None):
    """LayerNorm(x W^T + bias [+ residual]) in one kernel (fbbev_rows_linear_x3_ln); x (R, I), residual (R, O) rows, out (R, O)."""
    R, I = x.shape
    if x.stride(1) != 1 or (residual is not None and (tuple(residual.shape) != (R, out_features) or residual.stride(1) != 1)):
        raise FbbevError('rows_linear_x3_ln: rows must have unit column stride; residual must be (rows, out_features)')
    if out is None:
        out = torch.empty((R, out_features), dtype=F32, device=x.device)
    b = _dev(bias, F32, 'bias') if bias is not None else None
    with _on(x):
        _check(lib().fbbev_rows_linear_x3_ln(
            _dev(x, F32, 'x', contiguous=False), x.stride(0), fragments.data_ptr(), b, R, I, out_features,
            _dev(residual, F32, 'residual', contiguous=False) if residual is not None else None,
            residual.stride(0) if residual is not None else 0, _dev(ln_weight, F32, 'ln_weight'), _dev(ln_bias, F32, 'ln_bias'),
            float(eps), _dev(out, F32, 'out', contiguous=False), out.stride(0), _stream()), 'fbbev_rows_linear_x3_ln')
    return out


def rows_ffn_x3(x, w1_fragments, b1, w2_fragments, b2, hidden, out_features, residual=None, ln_weight=None, ln_bias=None, eps=1e-5):
    """[LayerNorm](W2 relu(W1 x + b1) + b2 [+ residual]) in one kernel (fbbev_rows_ffn_x3); x (R, I) rows, out (R, O)."""
    R, I = x.shape
    if x.stride(1) != 1 or (residual is not None and (tuple(residual.shape) != (R, out_features) or residual.stride(1) != 1)):
        raise FbbevError('rows_ffn_x3: rows must have unit column stride; residual must be (rows, out_features)')
    out = torch.empty((R, out_features), dtype=F32, device=x.device)
    with _on(x):
        _check(lib().fbbev_rows_ffn_x3(
            _dev(x, F32, 'x', contiguous=False), x.stride(0), w1_fragments.data_ptr(), _dev(b1, F32, 'b1'), w2_fragments.data_ptr(),
            _dev(b2, F32, 'b2'), R, I, int(hidden), int(out_features),
            _dev(residual, F32, 'residual', contiguous=False) if residual is not None else None,
            residual.stride(0) if residual is not None else 0,
            _dev(ln_weight, F32, 'ln_weight') if ln_weight is not None else None,
            _dev(ln_bias, F32, 'ln_bias') if ln_bias is not None else None, float(eps), _dev(out, F32, 'out'), out.stride(0), _stream()),
            'fbbev_rows_ffn_x3')
    return out


def rows_tail_ffn_x3_supported(x, residual0, embed, hidden):
    """shapes / alignment fbbev_rows_tail_ffn_x3 takes (probe the pointers too, a storage-offset view takes the fallback)"""
    def ok(t):
        return (t.is_cuda and t.dtype == F32 and t.dim() == 2 and t.shape[1] == embed and t.stride(1) == 1 and t.stride(0) % 4 == 0 and
                t.data_ptr() % 16 == 0)
    return embed % 16 == 0 and embed <= 80 and hidden % 64 == 0 and ok(x) and (residual0 is None or (ok(residual0) and residual0.shape == x.shape))


def rows_tail_ffn_x3(x, w0_fragments, b0, residual0, ln0_weight, ln0_bias, ln0_eps, w1_fragments, b1, w2_fragments, b2, hidden,
                     ln1_weight, ln1_bias, ln1_eps, tokens_per_image=None):
    """LayerNorm1(y1 + W2 relu(W1 y1 + b1) + b2) with y1 = LayerNorm0(x W0^T + b0 [+ residual0]) in one kernel
    (fbbev_rows_tail_ffn_x3): x (R, E) attention slots, residual0 (R, E) rows; out (R, E).  tokens_per_image = S: out is written
    as planes (R / S, E, S) instead (fbbev_rows_tail_ffn_x3_planes: the same values, the (B, C, Y, X) layout of the refined BEV)."""
    R, E = x.shape
    if tokens_per_image:
        S = int(tokens_per_image)
        if R % S:
            raise FbbevError('rows_tail_ffn_x3: rows must cover whole images')
        out = torch.empty((R // S, E, S), dtype=F32, device=x.device)
        with _on(x):
            _check(lib().fbbev_rows_tail_ffn_x3_planes(
                _dev(x, F32, 'x', contiguous=False), x.stride(0), w0_fragments.data_ptr(), _dev(b0, F32, 'b0'),
                _dev(residual0, F32, 'residual0', contiguous=False) if residual0 is not None else None,
                residual0.stride(0) if residual0 is not None else 0, _dev(ln0_weight, F32, 'ln0_weight'), _dev(ln0_bias, F32, 'ln0_bias'),
                float(ln0_eps), w1_fragments.data_ptr(), _dev(b1, F32, 'b1'), w2_fragments.data_ptr(), _dev(b2, F32, 'b2'), R, E, int(hidden),
                _dev(ln1_weight, F32, 'ln1_weight'), _dev(ln1_bias, F32, 'ln1_bias'), float(ln1_eps), S, _dev(out, F32, 'out'), _stream()),
                'fbbev_rows_tail_ffn_x3_planes')
        return out
    out = torch.empty((R, E), dtype=F32, device=x.device)
    with _on(x):
        _check(lib().fbbev_rows_tail_ffn_x3(
            _dev(x, F32, 'x', contiguous=False), x.stride(0), w0_fragments.data_ptr(), _dev(b0, F32, 'b0'),
            _dev(residual0, F32, 'residual0', contiguous=False) if residual0 is not None else None,
            residual0.stride(0) if residual0 is not None else 0, _dev(ln0_weight, F32, 'ln0_weight'), _dev(ln0_bias, F32, 'ln0_bias'),
            float(ln0_eps), w1_fragments.data_ptr(), _dev(b1, F32, 'b1'), w2_fragments.data_ptr(), _dev(b2, F32, 'b2'), R, E, int(hidden),
            _dev(ln1_weight, F32, 'ln1_weight'), _dev(ln1_bias, F32, 'ln1_bias'), float(ln1_eps), _dev(out, F32, 'out'), out.stride(0),
            _stream()), 'fbbev_rows_tail_ffn_x3')
    return out


def rows_linear_x3_planes(x, fragments, bias, tokens_per_image, heads, head_dim, out=None, dtype=None):
    """x (R, I) f32 rows of camera tokens (R = images * tokens_per_image) -> head planes (images, heads, tokens_per_image, head_dim)
    = value_proj written in the layout fbbev_da_cross_attn_fused samples (fbbev_rows_linear_x3_planes).  dtype bfloat16 / float16:
    the planes stored in 16 bits (fbbev_rows_linear_x3_planes_e: the fp32 result rounded once)."""
    R, I = x.shape
    if x.stride(1) != 1 or R % tokens_per_image != 0:
        raise FbbevError('rows_linear_x3_planes: rows must have unit column stride and cover whole images')
    shape = (R // tokens_per_image, heads, tokens_per_image, head_dim)
    dt = F32 if dtype is None else dtype
    if out is None:
        out = torch.empty(shape, dtype=dt, device=x.device)
    if tuple(out.shape) != shape or not out.is_contiguous() or out.dtype not in (F32, torch.bfloat16, torch.float16):
        raise FbbevError('rows_linear_x3_planes: out must be contiguous (images, heads, tokens, head_dim), f32 / bf16 / f16')
    b = _dev(bias, F32, 'bias') if bias is not None else None
    if out.dtype != F32:
        with _on(x):
            _check(lib().fbbev_rows_linear_x3_planes_e(_dev(x, F32, 'x', contiguous=False), x.stride(0), fragments.data_ptr(), b, R, I,
                                                       heads * head_dim, tokens_per_image, head_dim,
                                                       1 if out.dtype == torch.bfloat16 else 2, c_void_p(out.data_ptr()), _stream()),
                   'fbbev_rows_linear_x3_planes_e')
        return out
    with _on(x):
        _check(lib().fbbev_rows_linear_x3_planes(_dev(x, F32, 'x', contiguous=False), x.stride(0), fragments.data_ptr(), b, R, I,
                                                 heads * head_dim, tokens_per_image, head_dim, _dev(out, F32, 'out'), _stream()),
               'fbbev_rows_linear_x3_planes')
    return out


def rows_to_head_planes(rows, tokens_per_image, heads, head_dim):
    """(images * tokens_per_image, heads * head_dim) row-major camera tokens -> (images, heads, tokens_per_image, head_dim)"""
    R = rows.shape[0]
    out = torch.empty((R // tokens_per_image, heads, tokens_per_image, head_dim), dtype=F32, device=rows.device)
    with _on(rows):
        _check(lib().fbbev_rows_to_head_planes(_dev(rows, F32, 'rows'), R, tokens_per_image, heads, head_dim, _dev(out, F32, 'out'),
                                               _stream()), 'fbbev_rows_to_head_planes')
    return out


def volume_zreduce_supported(vol_view):
    """vol_view: (B, C, Y, X, Z) VIEW of a (B, C, Z, Y, X)-contiguous fp32 CUDA volume with Y*X % 4 == 0?"""
    return (vol_view.is_cuda and vol_view.dtype == F32 and vol_view.dim() == 5 and vol_view.permute(0, 1, 4, 2, 3).is_contiguous() and
            (vol_view.shape[2] * vol_view.shape[3]) % 4 == 0 and vol_view.data_ptr() % 16 == 0)


def volume_zreduce(vol_view, divisor):
    """(B, C, Y, X) = sum over Z of the (B, C, Y, X, Z) view / divisor (fbbev_volume_zreduce)."""
    B, C, Y, X, Z = vol_view.shape
    out = torch.empty((B, C, Y, X), dtype=F32, device=vol_view.device)
    with _on(vol_view):
        _check(lib().fbbev_volume_zreduce(vol_view.data_ptr(), B * C, Z, Y * X, float(divisor), _dev(out, F32, 'out'), _stream()),
               'fbbev_volume_zreduce')
    return out


def volume_zlast_supported(t):
    """t: a CONTIGUOUS (B, C, Y, X, Z) fp32 CUDA tensor with Z % 4 == 0 (a gradient handed over in the module's output shape)?"""
    return t.is_cuda and t.dtype == F32 and t.dim() == 5 and t.is_contiguous() and t.shape[-1] % 4 == 0 and t.data_ptr() % 16 == 0


def volume_zreduce_inner(t, divisor):
    """(B, C, Y, X) = sum over the innermost Z of a contiguous (B, C, Y, X, Z) tensor / divisor"""
    B, C, Y, X, Z = t.shape
    out = torch.empty((B, C, Y, X), dtype=F32, device=t.device)
    with _on(t):
        _check(lib().fbbev_volume_zreduce_inner(t.data_ptr(), B * C * Y * X, Z, float(divisor), _dev(out, F32, 'out'), _stream()),
               'fbbev_volume_zreduce_inner')
    return out


def volume_z_to_front(t):
    """contiguous (B, C, Y, X, Z) -> contiguous (B, C, Z, Y, X), the same elements"""
    B, C, Y, X, Z = t.shape
    out = torch.empty((B, C, Z, Y, X), dtype=F32, device=t.device)
    with _on(t):
        _check(lib().fbbev_volume_z_to_front(t.data_ptr(), B * C, Z, Y * X, _dev(out, F32, 'out'), _stream()), 'fbbev_volume_z_to_front')
    return out


def value_rows_to_head_planes(value, head_dim=None, interleaved=False):
    """(B*Ncam, S, M, HS) value rows (head m at m*HS, or chunk-major (HS/4, M, 4) when `interleaved`) -> (B*Ncam, M, S, Dh) planes"""
    BN, S, M, HS = value.shape
    Dh = HS if head_dim is None else int(head_dim)
    out = torch.empty((BN, M, S, Dh), dtype=F32, device=value.device)
    with _on(value):
        _check(lib().fbbev_value_rows_to_head_planes(_dev(value, F32, 'value'), BN * S, S, M, Dh, HS, 1 if interleaved else 0,
                                                     _dev(out, F32, 'planes'), _stream()), 'fbbev_value_rows_to_head_planes')
    return out


def da_cross_attn_fwd_planes_supported(B, Ncam, S, M, Dh, L, Q, P, Za):
    return bool(lib().fbbev_da_cross_attn_fwd_planes_supported(B, Ncam, S, M, Dh, L, Q, P, Za))


def da_cross_attn_fwd_planes(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, d0, dstep,
                             slots, head_minor=0, bev_w=0, min_level_width=2):
    """fbbev_da_cross_attn_fwd_planes: the training forward on head planes (offsets / softmaxed weights handed in)."""
    Ncam, B, Q, Za = mask.shape
    BN, M, S, Dh = planes.shape
    head_minor = int(head_minor)
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    with _on(planes):
        _check(lib().fbbev_da_cross_attn_fwd_planes(
            _dev(planes, F32, 'planes'), _dev(spatial_shapes, I64, 'spatial_shapes'), _dev(level_start_index, I64, 'level_start_index'),
            _dev(pred_depth, F32, 'pred_depth'), _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'),
            _dev(qdepth, F32, 'qdepth'), _dev(offsets, F32, 'offsets'), _dev(attn, F32, 'attn'), B, Ncam, S, M, Dh, L, Q, P, Za,
            pred_depth.shape[1], float(d0), float(dstep), head_minor & 3, int(bev_w or 0), int(min_level_width),
            _dev(slots, F32, 'slots'), _stream()), 'fbbev_da_cross_attn_fwd_planes')


def da_cross_attn_fused_supported(B, Ncam, S, M, Dh, L, Q, P, Za, bev_w):
    return bool(lib().fbbev_da_cross_attn_fused_supported(B, Ncam, S, M, Dh, L, Q, P, Za, int(bev_w)))


def da_cross_attn_fused(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, query, addend,
                        offsets_fragments, offsets_bias, attn_fragments, attn_bias, num_points, d0, dstep, bev_w, min_level_width,
                        slots, out_proj=None):
    """fbbev_da_cross_attn_fused: planes (B*Ncam, M, S, Dh) head-plane camera tokens; query (B, Q, E) rows [+ addend (P_, E) rows with
    B*Q % P_ == 0, e.g. the (Q, E) positional table]; fragments / biases of sampling_offsets and attention_weights in the module's
    row order (rows_linear_x3_fragments); slots (B, Q, E) written.  min_level_width: host value of the narrowest level's width."""
    Ncam, B, Q, Za = mask.shape
    BN, M, S, Dh = planes.shape
    E = M * Dh
    L = spatial_shapes.shape[0]
    DC = pred_depth.shape[1]
    if tuple(query.shape) != (B, Q, E) or query.stride(2) != 1 or query.stride(0) != Q * query.stride(1):
        raise FbbevError('da_cross_attn_fused: query must be (B, Q, E) rows with one row stride')
    if tuple(slots.shape) != (B, Q, E) or not slots.is_contiguous():
        raise FbbevError('da_cross_attn_fused: slots must be contiguous (B, Q, M*Dh)')
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    a_ptr, a_ld, a_per = None, 0, 1
    if addend is not None:
        if addend.dim() != 2 or addend.shape[1] != E or addend.stride(1) != 1 or (B * Q) % addend.shape[0] != 0:
            raise FbbevError('da_cross_attn_fused: addend must be (P, E) rows with B*Q % P == 0')
        a_ptr, a_ld, a_per = _dev(addend, F32, 'addend', contiguous=False), addend.stride(0), addend.shape[0]
    if out_proj is not None:      # (fragments, bias, residual (B, Q, E) or None, ln_weight, ln_bias, eps): the block's tail in the same workgroups
        wf, wb, res, lnw, lnb, eps = out_proj
        if res is not None and (tuple(res.shape) != (B, Q, E) or not res.is_contiguous()):
            raise FbbevError('da_cross_attn_fused: residual must be contiguous (B, Q, E)')
        with _on(planes):
            _check(lib().fbbev_da_cross_attn_fused_ln(
                _dev(planes, F32, 'planes'), _dev(spatial_shapes, I64, 'spatial_shapes'), _dev(level_start_index, I64, 'level_start_index'),
                _dev(pred_depth, F32, 'pred_depth'), _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'),
                _dev(qdepth, F32, 'qdepth'), _dev(query, F32, 'query', contiguous=False), query.stride(1), a_ptr, a_ld, a_per,
                offsets_fragments.data_ptr(), _dev(offsets_bias, F32, 'offsets_bias'), attn_fragments.data_ptr(),
                _dev(attn_bias, F32, 'attn_bias'), wf.data_ptr(), _dev(wb, F32, 'out_bias'),
                None if res is None else _dev(res, F32, 'residual'), E, _dev(lnw, F32, 'ln_weight'), _dev(lnb, F32, 'ln_bias'), float(eps),
                B, Ncam, S, M, Dh, L, Q, int(num_points), Za, DC, float(d0), float(dstep), int(bev_w), int(min_level_width),
                _dev(slots, F32, 'out'), _stream()), 'fbbev_da_cross_attn_fused_ln')
        return slots
    if planes.dtype in (torch.bfloat16, torch.float16):       # 16-bit camera tokens (storage option; products and sums stay fp32)
        if not planes.is_cuda or not planes.is_contiguous():
            raise FbbevError('da_cross_attn_fused: 16-bit planes must be contiguous GPU tensors')
        with _on(planes):
            _check(lib().fbbev_da_cross_attn_fused_e(
                c_void_p(planes.data_ptr()), 1 if planes.dtype == torch.bfloat16 else 2, _dev(spatial_shapes, I64, 'spatial_shapes'),
                _dev(level_start_index, I64, 'level_start_index'), _dev(pred_depth, F32, 'pred_depth'), _dev(ref_cam, F32, 'ref_cam'),
                _dev(mask, torch.uint8, 'mask'), _dev(qdepth, F32, 'qdepth'), _dev(query, F32, 'query', contiguous=False), query.stride(1),
                a_ptr, a_ld, a_per, offsets_fragments.data_ptr(), _dev(offsets_bias, F32, 'offsets_bias'), attn_fragments.data_ptr(),
                _dev(attn_bias, F32, 'attn_bias'), B, Ncam, S, M, Dh, L, Q, int(num_points), Za, DC, float(d0), float(dstep), int(bev_w),
                int(min_level_width), _dev(slots, F32, 'slots'), _stream()), 'fbbev_da_cross_attn_fused_e')
        return slots
    with _on(planes):
        _check(lib().fbbev_da_cross_attn_fused(
            _dev(planes, F32, 'planes'), _dev(spatial_shapes, I64, 'spatial_shapes'), _dev(level_start_index, I64, 'level_start_index'),
            _dev(pred_depth, F32, 'pred_depth'), _dev(ref_cam, F32, 'ref_cam'), _dev(mask, torch.uint8, 'mask'),
            _dev(qdepth, F32, 'qdepth'), _dev(query, F32, 'query', contiguous=False), query.stride(1), a_ptr, a_ld, a_per,
            offsets_fragments.data_ptr(), _dev(offsets_bias, F32, 'offsets_bias'), attn_fragments.data_ptr(),
            _dev(attn_bias, F32, 'attn_bias'), B, Ncam, S, M, Dh, L, Q, int(num_points), Za, DC, float(d0), float(dstep), int(bev_w),
            int(min_level_width), _dev(slots, F32, 'slots'), _stream()), 'fbbev_da_cross_attn_fused')
    return slots


def msda_self_fused_supported(B, S, M, Dh, L, Q, P, bev_w):
    return bool(lib().fbbev_msda_self_fused_supported(B, S, M, Dh, L, Q, P, int(bev_w)))


def msda_self_fused(planes, reference_points, query, addend, offsets_fragments, offsets_bias, attn_fragments, attn_bias, num_points,
                    bev_w, level_hw, out, out_proj=None):
    """fbbev_msda_self_fused: planes (B, M, S, Dh) head-plane value tokens; reference_points (B, Q, 1, 2); query (B, Q, E) rows
    [+ addend (P_, E) rows]; fragments / biases of sampling_offsets and attention_weights in the module's row order; out (B, Q, E).
    out_proj = (fragments, bias, residual (B, Q, E) or None, ln_weight, ln_bias, eps): the block's tail in the same workgroups
    (fbbev_msda_self_fused_ln): out = LayerNorm(output_proj(attention) + residual)."""
    B, M, S, Dh = planes.shape
    Q = query.shape[1]
    E = M * Dh
    if tuple(query.shape) != (B, Q, E) or query.stride(2) != 1 or query.stride(0) != Q * query.stride(1):
        raise FbbevError('msda_self_fused: query must be (B, Q, E) rows with one row stride')
    if tuple(out.shape) != (B, Q, E) or not out.is_contiguous():
        raise FbbevError('msda_self_fused: out must be contiguous (B, Q, M*Dh)')
    if tuple(reference_points.shape) != (B, Q, 1, 2):
        raise FbbevError('msda_self_fused: reference_points must be (B, Q, 1, 2)')
    a_ptr, a_ld, a_per = None, 0, 1
    if addend is not None:
        if addend.dim() != 2 or addend.shape[1] != E or addend.stride(1) != 1 or (B * Q) % addend.shape[0] != 0:
            raise FbbevError('msda_self_fused: addend must be (P, E) rows with B*Q % P == 0')
        a_ptr, a_ld, a_per = _dev(addend, F32, 'addend', contiguous=False), addend.stride(0), addend.shape[0]
    if out_proj is not None:
        wf, wb, res, lnw, lnb, eps = out_proj
        if res is not None and (tuple(res.shape) != (B, Q, E) or not res.is_contiguous()):
            raise FbbevError('msda_self_fused: residual must be contiguous (B, Q, E)')
        with _on(planes):
            _check(lib().fbbev_msda_self_fused_ln(
                _dev(planes, F32, 'planes'), _dev(reference_points, F32, 'reference_points'), _dev(query, F32, 'query', contiguous=False),
                query.stride(1), a_ptr, a_ld, a_per, offsets_fragments.data_ptr(), _dev(offsets_bias, F32, 'offsets_bias'),
                attn_fragments.data_ptr(), _dev(attn_bias, F32, 'attn_bias'), wf.data_ptr(), _dev(wb, F32, 'out_bias'),
                None if res is None else _dev(res, F32, 'residual'), E, _dev(lnw, F32, 'ln_weight'), _dev(lnb, F32, 'ln_bias'),
                float(eps), B, S, M, Dh, 1, Q, int(num_points), int(bev_w), int(level_hw[0]), int(level_hw[1]), _dev(out, F32, 'out'),
                _stream()), 'fbbev_msda_self_fused_ln')
        return out
    with _on(planes):
        _check(lib().fbbev_msda_self_fused(
            _dev(planes, F32, 'planes'), _dev(reference_points, F32, 'reference_points'), _dev(query, F32, 'query', contiguous=False),
            query.stride(1), a_ptr, a_ld, a_per, offsets_fragments.data_ptr(), _dev(offsets_bias, F32, 'offsets_bias'),
            attn_fragments.data_ptr(), _dev(attn_bias, F32, 'attn_bias'), B, S, M, Dh, 1, Q, int(num_points), int(bev_w),
            int(level_hw[0]), int(level_hw[1]), _dev(out, F32, 'out'), _stream()), 'fbbev_msda_self_fused')
    return out


def layernorm_bwd(x, grad_out, weight, eps):
    """Backward of `layernorm` (no residual): returns (grad_x, grad_weight, grad_bias); the parameter gradients are the sum
    of the kernel's per-workgroup partial rows."""
    C = x.shape[-1]
    rows = x.numel() // C
    n = lib().fbbev_layernorm_bwd_partials(rows)
    partial = torch.empty((n, 2, C), dtype=F32, device=x.device)
    grad_x = torch.empty_like(x)
    with _on(x):
        _check(lib().fbbev_layernorm_bwd(_dev(x, F32, 'x'), _dev(grad_out, F32, 'grad_out'), _dev(weight, F32, 'weight'),
                                         float(eps), rows, C, _dev(grad_x, F32, 'grad_x'), _dev(partial, F32, 'partial'),
                                         _stream()), 'fbbev_layernorm_bwd')
    gwb = torch.empty((2, C), dtype=F32, device=x.device)
    with _on(x):          # fixed-order sum of the partial rows (ATen's dim-0 reduction of this shape is one latency chain per column)
        _check(lib().fbbev_sum_partials(_dev(partial, F32, 'partial'), n, 2 * C, _dev(gwb, F32, 'out'), _stream()), 'fbbev_sum_partials')
    return grad_x, gwb[0], gwb[1]


def softmax_groups(x, group, out=None):
    """softmax over groups of `group` consecutive floats of a contiguous f32 tensor (fbbev_softmax_groups); out may be x"""
    out = torch.empty_like(x) if out is None else out
    with _on(x):
        _check(lib().fbbev_softmax_groups(_dev(x, F32, 'x'), x.numel() // group, int(group), _dev(out, F32, 'out'), _stream()), 'fbbev_softmax_groups')
    return out


def softmax_groups_bwd(y, grad_y, group, out=None):
    """grad_x = y * (grad_y - sum over the group of y * grad_y) (fbbev_softmax_groups_bwd); out may be grad_y"""
    out = torch.empty_like(grad_y) if out is None else out
    with _on(y):
        _check(lib().fbbev_softmax_groups_bwd(_dev(y, F32, 'y'), _dev(grad_y, F32, 'grad_y'), y.numel() // group, int(group),
                                              _dev(out, F32, 'out'), _stream()), 'fbbev_softmax_groups_bwd')
    return out


def touch(*tensors):
    """fbbev_touch: read up to 8 device tensors once (a read-ahead of a later kernel's gather sources); values unused"""
    ts = [t for t in tensors if t is not None and t.numel() > 0][:8]
    if not ts:
        return
    ptrs = (c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    nbytes = (c_size_t * len(ts))(*[t.numel() * t.element_size() for t in ts])
    with _on(ts[0]):
        _check(lib().fbbev_touch(ptrs, nbytes, len(ts), _stream()), 'fbbev_touch')


def rows_linear_x3_train(x, fragments, bias, out_features, relu=False, addend=None, residual=None, mask=None, out=None):
    """fbbev_rows_linear_x3_train: out = ((x [+ addend[r % P]]) W^T + bias) [ReLU]) * [mask > 0] + residual; residual may be `out`."""
    R, I = x.shape
    if x.stride(1) != 1:
        raise FbbevError('rows_linear_x3_train: rows must have unit column stride')
    if out is None:
        out = torch.empty((R, out_features), dtype=F32, device=x.device)
    for t, nm in ((residual, 'residual'), (mask, 'mask'), (out, 'out')):
        if t is not None and (tuple(t.shape) != (R, out_features) or t.stride(1) != 1):
            raise FbbevError(f'rows_linear_x3_train: {nm} must be (rows, out_features) with unit column stride')
    a_ptr, a_ld, a_per = None, 0, 1
    if addend is not None:
        if addend.dim() != 2 or addend.shape[1] != I or addend.stride(1) != 1 or R % addend.shape[0] != 0:
            raise FbbevError('rows_linear_x3_train: addend must be (P, in_features) rows with rows % P == 0')
        a_ptr, a_ld, a_per = _dev(addend, F32, 'addend', contiguous=False), addend.stride(0), addend.shape[0]
    with _on(x):
        _check(lib().fbbev_rows_linear_x3_train(
            _dev(x, F32, 'x', contiguous=False), x.stride(0), a_ptr, a_ld, a_per, fragments.data_ptr(),
            _dev(bias, F32, 'bias') if bias is not None else None, R, I, out_features, 1 if relu else 0,
            _dev(residual, F32, 'residual', contiguous=False) if residual is not None else None,
            residual.stride(0) if residual is not None else 0, _dev(mask, F32, 'mask', contiguous=False) if mask is not None else None,
            mask.stride(0) if mask is not None else 0, _dev(out, F32, 'out', contiguous=False), out.stride(0), _stream()),
            'fbbev_rows_linear_x3_train')
    return out


def sum_leading(x, x2=None):
    """(B, ...) f32 contiguous [+ x2 of the same shape] -> sum over the leading dimension, ascending b (fbbev_sum_leading)"""
    B = x.shape[0]
    N = x.numel() // B
    out = torch.empty(x.shape[1:], dtype=F32, device=x.device)
    with _on(x):
        _check(lib().fbbev_sum_leading(_dev(x, F32, 'x'), _dev(x2, F32, 'x2') if x2 is not None else None, B, N, _dev(out, F32, 'out'),
                                       _stream()), 'fbbev_sum_leading')
    return out


def rows_wgrad_x3(grad_out, x, bias=True, addend=None):
    """grad_out (R, O), x (R, I) f32 rows (unit column stride) -> (grad_weight (O, I), grad_bias (O) or None): fbbev_rows_wgrad_x3,
    split-operand MFMA over the rows, fixed-order reduction of the per-workgroup partial results (bit-stable)."""
    R, O = grad_out.shape
    I = x.shape[1]
    if x.shape[0] != R or grad_out.stride(1) != 1 or x.stride(1) != 1:
        raise FbbevError('rows_wgrad_x3: grad_out (R, O) and x (R, I) rows with unit column stride')
    need = lib().fbbev_rows_wgrad_x3_ws_bytes(R, I, O)
    if need == 0:
        raise FbbevError('rows_wgrad_x3: unsupported shape')
    ws = torch.empty(need // 4, dtype=F32, device=x.device)
    gw = torch.empty((O, I), dtype=F32, device=x.device)
    gb = torch.empty((O,), dtype=F32, device=x.device) if bias else None
    a_ptr, a_ld, a_per = None, 0, 1
    if addend is not None:      # the layer's input rows were x[r] + addend[r % P]
        if addend.dim() != 2 or addend.shape[1] != I or addend.stride(1) != 1 or R % addend.shape[0] != 0:
            raise FbbevError('rows_wgrad_x3: addend must be (P, in_features) rows with rows % P == 0')
        a_ptr, a_ld, a_per = _dev(addend, F32, 'addend', contiguous=False), addend.stride(0), addend.shape[0]
    with _on(x):
        _check(lib().fbbev_rows_wgrad_x3(_dev(grad_out, F32, 'grad_out', contiguous=False), grad_out.stride(0),
                                         _dev(x, F32, 'x', contiguous=False), x.stride(0), a_ptr, a_ld, a_per, R, I, O, _dev(gw, F32, 'grad_weight'),
                                         _dev(gb, F32, 'grad_bias') if gb is not None else None, c_void_p(ws.data_ptr()), need, _stream()),
               'fbbev_rows_wgrad_x3')
    return gw, gb


def rows_wgrad_x3_supported(grad_out, x):
    return (grad_out.is_cuda and grad_out.dtype == F32 and x.dtype == F32 and grad_out.dim() == 2 and x.dim() == 2 and
            grad_out.stride(1) == 1 and x.stride(1) == 1 and grad_out.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and
            grad_out.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and grad_out.shape[1] % 4 == 0 and x.shape[1] % 4 == 0)


def conv3d_ndhwc(x, weight_fragments, bias, out, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None, transposed=False):
    """x (B,Di,Hi,Wi,Cin) f32 contiguous (NDHWC); out (B,Do,Ho,Wo,Cout) [transposed: (B,2Di,2Hi,2Wi,Cout)] contiguous;
    weight_fragments / bias as built by fb_bev_amd.mfma_conv3d (batch norm folded, MFMA A-fragment order)."""
    B, Di, Hi, Wi, Cin = x.shape
    if transposed:
        Do, Ho, Wo = Di, Hi, Wi
        want = (B, 2 * Di, 2 * Hi, 2 * Wi, Cout)
    else:
        Do, Ho, Wo = [(n + 2 * pad - ksize) // stride + 1 for n in (Di, Hi, Wi)]
        want = (B, Do, Ho, Wo, Cout)
    if tuple(out.shape) != want or (residual is not None and tuple(residual.shape) != want):
        raise FbbevError(f'conv3d_ndhwc: out / residual must be {want}')
    with _on(x):
        _check(lib().fbbev_conv3d_ndhwc(
            _dev(x, F32, 'x'), _dev(weight_fragments, F32, 'weight_fragments'), _dev(bias, F32, 'bias'),
            None if residual is None else _dev(residual, F32, 'residual'), B, Di, Hi, Wi, Cin, Do, Ho, Wo, int(Cout), int(ksize),
            int(stride), int(pad), 1 if relu else 0, 1 if transposed else 0, _dev(out, F32, 'out'), _stream()),
            'fbbev_conv3d_ndhwc')
    return out


def conv2d_nhwc(x, weight_fragments, bias, out, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None):
    """x (B,H,W,Cin) f32 contiguous (NHWC); out (B,Ho,Wo,Cout); weights as mfma_conv3d.weight_fragments(w[:, :, None])."""
    B, Hi, Wi, Cin = x.shape
    Ho, Wo = [(n + 2 * pad - ksize) // stride + 1 for n in (Hi, Wi)]
    if tuple(out.shape) != (B, Ho, Wo, Cout) or (residual is not None and tuple(residual.shape) != (B, Ho, Wo, Cout)):
        raise FbbevError(f'conv2d_nhwc: out / residual must be {(B, Ho, Wo, Cout)}')
    with _on(x):
        _check(lib().fbbev_conv2d_nhwc(
            _dev(x, F32, 'x'), _dev(weight_fragments, F32, 'weight_fragments'), _dev(bias, F32, 'bias'),
            None if residual is None else _dev(residual, F32, 'residual'), B, Hi, Wi, Cin, Ho, Wo, int(Cout), int(ksize), int(stride),
            int(pad), 1 if relu else 0, _dev(out, F32, 'out'), _stream()), 'fbbev_conv2d_nhwc')
    return out


def conv3d_ndhwc_bf16(x, weight_fragments_bf16, bias, out, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None,
                      transposed=False, planar=False):
    """bf16-MFMA variant of conv3d_ndhwc (planar=True: x (B,1,H,W,Cin), the 2-D case); weights from
    mfma_conv3d.weight_fragments_bf16 (torch.bfloat16 tensor)."""
    B, Di, Hi, Wi, Cin = x.shape
    Do, Ho, Wo = out.shape[1:4]
    if transposed:
        Do, Ho, Wo = Di, Hi, Wi
    if weight_fragments_bf16.dtype != torch.bfloat16:
        raise FbbevError('weight_fragments_bf16 must be a bfloat16 tensor')
    with _on(x):
        _check(lib().fbbev_conv3d_ndhwc_bf16(
            _dev(x, F32, 'x'), _dev(weight_fragments_bf16, torch.bfloat16, 'weight_fragments_bf16'), _dev(bias, F32, 'bias'),
            None if residual is None else _dev(residual, F32, 'residual'), B, Di, Hi, Wi, Cin, int(Do), int(Ho), int(Wo), int(Cout),
            int(ksize), int(stride), int(pad), 1 if relu else 0, 1 if transposed else 0, 1 if planar else 0,
            _dev(out, F32, 'out'), _stream()), 'fbbev_conv3d_ndhwc_bf16')
    return out


def conv3d_k3s1_tiled_bf16(x, weight_fragments_bf16, bias, out, Cout, relu=False, residual=None):
    """3x3x3 / stride 1 / padding 1 on (B,D,H,W,Cin) f32 with the LDS-staged halo tile (bf16 MFMA)."""
    B, D, H, W, Cin = x.shape
    if tuple(out.shape) != (B, D, H, W, Cout) or weight_fragments_bf16.dtype != torch.bfloat16:
        raise FbbevError('conv3d_k3s1_tiled_bf16: bad out shape / weight dtype')
    with _on(x):
        _check(lib().fbbev_conv3d_k3s1_tiled_bf16(
            _dev(x, F32, 'x'), _dev(weight_fragments_bf16, torch.bfloat16, 'weight_fragments_bf16'), _dev(bias, F32, 'bias'),
            None if residual is None else _dev(residual, F32, 'residual'), B, D, H, W, Cin, int(Cout), 1 if relu else 0,
            _dev(out, F32, 'out'), _stream()), 'fbbev_conv3d_k3s1_tiled_bf16')
    return out


def conv3d_dgrad_ndhwc(dy, weight_fragments_t, dx, ksize=3, stride=1, pad=1):
    """dx (B,Di,Hi,Wi,Cin) <- data gradient of the convolution whose output gradient is dy (B,Do,Ho,Wo,Cout);
    weight_fragments_t = mfma_conv3d.weight_fragments(weight.transpose(0, 1))."""
    B, Do, Ho, Wo, Cout = dy.shape
    _, Di, Hi, Wi, Cin = dx.shape
    zero = torch.zeros((Cin + 15) // 16 * 16, dtype=F32, device=dy.device)
    with _on(dy):
        _check(lib().fbbev_conv3d_dgrad_ndhwc(_dev(dy, F32, 'dy'), _dev(weight_fragments_t, F32, 'weight_fragments_t'),
                                              _dev(zero, F32, 'zero'), B, Do, Ho, Wo, Cout, Di, Hi, Wi, Cin, int(ksize), int(stride),
                                              int(pad), _dev(dx, F32, 'dx'), _stream()), 'fbbev_conv3d_dgrad_ndhwc')
    return dx


def conv3d_wgrad_ndhwc(x, dy, dw, ksize=3, stride=1, pad=1):
    """dw (ksize^3, Cout, Cin) f32, zero on entry, += sum_v dy[v] (x) x[v*stride + tap - pad]."""
    B, Di, Hi, Wi, Cin = x.shape
    _, Do, Ho, Wo, Cout = dy.shape
    if tuple(dw.shape) != (ksize ** 3, Cout, Cin):
        raise FbbevError('conv3d_wgrad_ndhwc: dw must be (ksize^3, Cout, Cin)')
    with _on(x):
        _check(lib().fbbev_conv3d_wgrad_ndhwc(_dev(x, F32, 'x'), _dev(dy, F32, 'dy'), B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout,
                                              int(ksize), int(stride), int(pad), _dev(dw, F32, 'dw'), _stream()),
               'fbbev_conv3d_wgrad_ndhwc')
    return dw


def blend_levels_ndhwc(level0, coarse, wsoft, out):
    """out = wsoft[...,0] * level0 + sum_k wsoft[...,k] * trilinear_upsample(coarse[k-1]); all NDHWC f32 contiguous."""
    import ctypes
    B, D, H, W, C = level0.shape
    n = len(coarse)
    ptrs = (c_void_p * max(n, 1))(*[_dev(t, F32, 'coarse').value for t in coarse]) if n else None
    dims = (c_int * max(3 * n, 1))(*[int(v) for t in coarse for v in t.shape[1:4]]) if n else None
    for t in coarse:
        if t.shape[0] != B or t.shape[4] != C:
            raise FbbevError('blend_levels_ndhwc: coarse level batch / channels differ')
    if tuple(out.shape) != tuple(level0.shape) or tuple(wsoft.shape[:4]) != (B, D, H, W):
        raise FbbevError('blend_levels_ndhwc: bad out / wsoft shape')
    with _on(level0):
        _check(lib().fbbev_blend_levels_ndhwc(
            _dev(level0, F32, 'level0'), ctypes.cast(ptrs, c_void_p) if n else None, ctypes.cast(dims, c_void_p) if n else None,
            n, _dev(wsoft, F32, 'wsoft'), int(wsoft.shape[4]), B, D, H, W, C, _dev(out, F32, 'out'), _stream()),
            'fbbev_blend_levels_ndhwc')
    return out


def history_fused_x3_vm(history, flow, nxt, grid_zyx, w1, bias1, w2, bias2, out):
    """fbbev_history_fused_x3_vm: warp + new ring + both split-operand convolutions in ONE launch.  history (B, T, N, C) and
    nxt (B, T+1, N, C) 16-bit voxel-major rings (nxt[:, 0] = the current frame, already stored), C = Cout = 80; writes nxt[:, 1:]
    (== history_warp_vm) and out (B, Cout, N) f32 (== history_conv(nxt, ..., compute='bf16x3'))."""
    Z, Y, X = grid_zyx
    B, T, N, C = history.shape
    Cout = w2.shape[0]
    if (tuple(nxt.shape) != (B, T + 1, N, C) or nxt.dtype != history.dtype or history.dtype not in (torch.bfloat16, torch.float16)
            or N != Z * Y * X or history.stride()[1:] != (N * C, C, 1) or nxt.stride()[1:] != (N * C, C, 1)
            or tuple(out.shape) != (B, Cout, N)):
        raise FbbevError('history_fused_x3_vm: bad history / next / out layout')
    ws = torch.empty((2 + T) * C * max(C, Cout, 96) + B * (T + 1) * C + 16, dtype=torch.float32, device=history.device)
    with _on(history):
        _check(lib().fbbev_history_fused_x3_vm(
            _dev(history, history.dtype, 'history', contiguous=False), history.stride(0),
            _dev(nxt, nxt.dtype, 'next', contiguous=False), nxt.stride(0), _dev(flow, F32, 'rt_flow'), _dev(w1, F32, 'w1'),
            _dev(bias1, F32, 'bias1'), _dev(w2, F32, 'w2'), _dev(bias2, F32, 'bias2'), B, T, C, Cout, Z, Y, X,
            _dev(out, F32, 'out'), c_void_p(ws.data_ptr()), ws.numel() * 4, ELEM_TYPE[history.dtype], _stream()),
            'fbbev_history_fused_x3_vm')
    return out


def history_step_x3_vm(history, flow, nxt, grid_zyx, w1, bias1, w2, bias2, out, chunks=0):
    """fbbev_history_step_x3_vm: history (B, T, N, C) 16-bit voxel-major, nxt (B, T+1, N, C) with the current frame in nxt[:, 0];
    writes nxt[:, 1:] (the warped frames) and out (B, Cout, N) f32 (both split-operand convolutions), as a two-stream pipeline
    over `chunks` bands of rows (0: default, 1: back to back)."""
    Z, Y, X = grid_zyx
    B, T, N, C = history.shape
    Cout = w2.shape[0]
    if (tuple(nxt.shape) != (B, T + 1, N, C) or nxt.dtype != history.dtype or history.dtype not in (torch.bfloat16, torch.float16)
            or N != Z * Y * X or history.stride()[1:] != (N * C, C, 1) or nxt.stride()[1:] != (N * C, C, 1)
            or tuple(out.shape) != (B, Cout, N)):
        raise FbbevError('history_step_x3_vm: bad history / next / out layout')
    ws = torch.empty((2 + T) * C * max(C, Cout, 96) + B * (T + 1) * C, dtype=torch.float32, device=history.device)
    with _on(history):
        _check(lib().fbbev_history_step_x3_vm(
            _dev(history, history.dtype, 'history', contiguous=False), history.stride(0),
            _dev(nxt, nxt.dtype, 'next', contiguous=False), nxt.stride(0), _dev(flow, F32, 'rt_flow'), _dev(w1, F32, 'w1'),
            _dev(bias1, F32, 'bias1'), _dev(w2, F32, 'w2'), _dev(bias2, F32, 'bias2'), B, T, C, Cout, Z, Y, X,
            _dev(out, F32, 'out'), c_void_p(ws.data_ptr()), ws.numel() * 4, ELEM_TYPE[history.dtype], int(chunks), _stream()),
            'fbbev_history_step_x3_vm')
    return out


def history_conv(feats, w1, bias1, w2, bias2, out, compute=torch.float32, voxel_major=False):
    """feats (B, T1*C, N) f32 / bf16 / f16 whose per-sample block is contiguous; w1 (C,C); bias1 (B*T1, C); w2 (Cout, T1*C);
    bias2 (Cout); out (B, Cout, N) f32 contiguous -> out = relu(bias2 + sum_t w2_t . relu(w1 . x_t + bias1_t)).
    compute=torch.bfloat16: both GEMMs on the bf16 MFMA with fp32 accumulation (fbbev_history_conv_bf16);
    compute='bf16x3': the same at fp32-grade precision, operands split into two bf16 terms (fbbev_history_conv_bf16x3);
    voxel_major=True: feats is (B, T1, N, C), the voxel-major ring (C = Cout in {16, 80})."""
    C = w1.shape[0]
    Cout = w2.shape[0]
    if voxel_major:
        B, T1, N, _ = feats.shape
        ok = feats.dim() == 4 and feats.shape[3] == C and feats.stride()[1:] == (N * C, C, 1)
    else:
        B, TC, N = feats.shape
        T1 = TC // C
        ok = feats.stride()[1:] == (N, 1)
    if not ok or tuple(out.shape) != (B, Cout, N) or feats.dtype not in ELEM_TYPE:
        raise FbbevError('history_conv: bad feats / out layout')
    if compute not in (torch.float32, torch.bfloat16, 'bf16x3'):
        raise FbbevError("history_conv: compute is float32, bfloat16 or 'bf16x3'")
    if compute == 'bf16x3' and not (voxel_major and feats.dtype in (torch.bfloat16, torch.float16)):
        raise FbbevError("history_conv: compute='bf16x3' needs a 16-bit voxel-major ring")
    # fragment-ordered weights (+ the scaled biases of the split-operand route)
    ws = torch.empty((1 + T1) * C * max(C, Cout, 96) + B * T1 * C, dtype=torch.float32, device=feats.device)
    args = (_dev(feats, feats.dtype, 'feats', contiguous=False), feats.stride(0), _dev(w1, F32, 'w1'),
            _dev(bias1, F32, 'bias1'), _dev(w2, F32, 'w2'), _dev(bias2, F32, 'bias2'),
            B, T1, C, Cout, N, _dev(out, F32, 'out'), c_void_p(ws.data_ptr()), ws.numel() * 4)
    with _on(feats):
        if compute == 'bf16x3':
            _check(lib().fbbev_history_conv_bf16x3(*args, ELEM_TYPE[feats.dtype], _stream()), 'fbbev_history_conv_bf16x3')
        elif compute == torch.bfloat16:
            _check(lib().fbbev_history_conv_bf16(*args, 1 if voxel_major else 0, ELEM_TYPE[feats.dtype], _stream()),
                   'fbbev_history_conv_bf16')
        elif voxel_major:
            _check(lib().fbbev_history_conv_vm(*args, ELEM_TYPE[feats.dtype], _stream()), 'fbbev_history_conv_vm')
        else:
            _check(lib().fbbev_history_conv_e(*args, ELEM_TYPE[feats.dtype], _stream()), 'fbbev_history_conv_e')
    return out
