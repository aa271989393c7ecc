"""Drop-in for `mmcv._ext.ms_deform_attn_forward / ms_deform_attn_backward` (mmcv-full 1.5.2,
loaded by the reference at bevformer_utils/multi_scale_deformable_attn_function.py:18-19) and
mirror of MultiScaleDeformableAttnFunction_fp32 (:99-172).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _capi

__all__ = ['ms_deform_attn_forward', 'ms_deform_attn_backward',
           'MultiScaleDeformableAttnFunction_fp32', 'MultiScaleDeformableAttnFunction_fp16']


def _meta(t, like):
    return t.to(device=like.device, dtype=torch.int64).contiguous()


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                           attention_weights, im2col_step=64):
    """-> (B, Q, M*Dh).  `im2col_step` is accepted for signature compatibility and ignored: any
    batch size works (mmcv asserts batch % min(batch, im2col_step) == 0)."""
    B, _, M, Dh = value.shape
    Q = sampling_locations.shape[1]
    out = value.new_empty((B, Q, M * Dh))
    _capi.msda_fwd(value, _meta(value_spatial_shapes, value), _meta(value_level_start_index, value),
                   sampling_locations, attention_weights, out)
    return out


def _host_level_hw(spatial_shapes):
    """Host (h, w) pairs of a `spatial_shapes` tensor built by backward_projection.const_tensor (it keeps the Python values it
    was made from in `_fbbev_host`); None for any other tensor -- never a device read."""
    hv = getattr(spatial_shapes, '_fbbev_host', None)
    if hv is None:
        return None
    try:
        return [(int(h), int(w)) for h, w in hv]
    except (TypeError, ValueError):
        return None


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                            attention_weights, grad_output, grad_value, grad_sampling_loc,
                            grad_attn_weight, im2col_step=64, level_hw=None):
    """In place into the three pre-zeroed grad tensors (multi_scale_deformable_attn_function.py:155-169).  level_hw (host
    (h, w) per level): the atomic-free, bit-reproducible kernels (fbbev_msda_bwd_ws) where they apply."""
    _capi.msda_bwd(value, _meta(value_spatial_shapes, value), _meta(value_level_start_index, value),
                   sampling_locations, attention_weights, grad_output, grad_value, grad_sampling_loc,
                   grad_attn_weight, level_hw=level_hw)


class MultiScaleDeformableAttnFunction_fp32(Function):
    """multi_scale_deformable_attn_function.py:99-172 (inputs cast to fp32, once-differentiable)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        value = value.float().contiguous()
        sampling_locations = sampling_locations.float().contiguous()
        attention_weights = attention_weights.float().contiguous()
        ctx.im2col_step = im2col_step
        ctx.level_hw = _host_level_hw(value_spatial_shapes)          # attributes do not survive save_for_backward
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights, im2col_step=im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        B, S, M, Dh = value.shape
        _, Q, _, L, P, _ = sampling_locations.shape
        written = value.is_cuda and _capi.msda_bwd_ws_bytes(B, S, M, Dh, L, Q, P, ctx.level_hw) > 0
        grad_value = torch.empty_like(value) if written else torch.zeros_like(value)
        grad_sampling_loc = torch.zeros_like(sampling_locations)
        grad_attn_weight = torch.zeros_like(attention_weights)
        ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_locations,
                                attention_weights, grad_output.float().contiguous(), grad_value,
                                grad_sampling_loc, grad_attn_weight, im2col_step=ctx.im2col_step,
                                level_hw=ctx.level_hw if written else None)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


# the reference selects the fp32 function in both branches (spatial_cross_attention_depth.py:580-583)
MultiScaleDeformableAttnFunction_fp16 = MultiScaleDeformableAttnFunction_fp32
