"""Backward projection (BEV -> image refinement with depth-aware deformable cross-attention) --
host-side mirror of mmdet3d/models/fbbev/view_transformation/backward_projection/:
  BackwardProjection ................. backward_projection.py:34-133
  BEVFormer .......................... bevformer_utils/bevformer.py:22-132
  bevformer_encoder .................. bevformer_utils/bevformer_encoder.py:27-203
  BEVFormerEncoderLayer .............. bevformer_utils/bevformer_encoder.py:206-377
  DA_SpatialCrossAttention ........... bevformer_utils/spatial_cross_attention_depth.py:31-223
  DA_MSDeformableAttention ........... bevformer_utils/spatial_cross_attention_depth.py:361-601
  CustormLearnedPositionalEncoding ... bevformer_utils/positional_encoding.py:11-68
  MultiScaleDeformableAttention, FFN . mmcv-full 1.5.2 classes the config names (cfg :176-198)
Class names, constructor arguments and state_dict keys follow the reference so the
`backward_projection=dict(type='BackwardProjection', ...)` block of
occupancy_configs/fb_occ/fbocc-r50-cbgs_depth_16f_16x4_20e.py:154-211 builds unchanged via `build()`
and reference checkpoints load.  All deformable sampling runs on the HIP kernels of libfbbev_hip.so:
  * training / autograd: composite path over ms_deform_attn_forward/backward (vectorised rebatch,
    one host sync for max_len -- the reference needs 6*B);
  * inference (no grad): fbbev_da_cross_attn_fwd, one fused launch, no sync, no padding.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi
from .ms_deform_attn import MultiScaleDeformableAttnFunction_fp32
from . import rows_linear as _RL
from .rows_linear import Linear, X3Weights, linear_rows, linear_x3, x3_ok

REGISTRY = {}


def register(cls):
    REGISTRY[cls.__name__] = cls
    return cls


def build(cfg, **extra):
    """mmcv-style build_from_cfg over one flat registry (the reference spreads these names over
    HEADS / TRANSFORMER / TRANSFORMER_LAYER_SEQUENCE / TRANSFORMER_LAYER / ATTENTION /
    POSITIONAL_ENCODING / FEEDFORWARD_NETWORK)."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    cfg.update(extra)
    if typ not in REGISTRY and typ.endswith('TRT') and typ[:-3] in REGISTRY:
        # the *TRT classes of the deployment config (fbocc-..._trt.py; backward_projection.py:137,
        # spatial_cross_attention_depth.py:227,604, multi_scale_deformable_attn_function.py:175) are the same modules
        # with the same constructor arguments and parameters, re-worded for TensorRT export: built as the native class
        typ = typ[:-3]
    return REGISTRY[typ](**cfg)


_CONST = {}


def const_tensor(values, device, dtype=torch.long):
    """Small constant device tensors (spatial shapes, level starts) built once per (value, device): creating
    them from Python lists on every forward is a pageable H2D copy, which also breaks hipGraph capture."""
    key = (repr(values), str(device), dtype)
    if key not in _CONST:
        t = torch.as_tensor(values, dtype=dtype, device=device)
        t._fbbev_host = values          # the Python values ride along: kernels that plan on the host need no device read
        _CONST[key] = t
    return _CONST[key]


# ---- inference prefetch (round 5): the part of the backward projection that does not depend on the lift-splat's Z-mean -- camera-token
# rows, their value projection as head planes, BEV -> image point sampling -- computed on a SIDE stream while the forward projection's
# latency-bound ranking chain and Z-mean run on the main one (FBViewTransform.forward); consumed by the forward that follows when it
# is called with the same input tensors.  FBBEV_BP_PREFETCH=0 turns it off (A/B knob); never active during hipGraph capture.
import os as _os0
PREFETCH = _os0.environ.get('FBBEV_BP_PREFETCH', '1') != '0'
PREFETCH_MIN_QUERIES = int(_os0.environ.get('FBBEV_BP_PREFETCH_MIN_QUERIES', '120000'))


class _Prefetch:
    def __init__(self):
        self.rows = self.rows_key = self.sampling = self.sampling_key = self.event = None
        self.planes = {}                  # id(cross-attention module) -> (data_ptr of the rows, head planes)


_PRE = None                               # set by BackwardProjection.forward around its transformer call


def _tkey(ts):
    return tuple((t.data_ptr(), tuple(t.shape), t._version) for t in ts)


_OUT_PLANES = None        # inference: {'tokens': Y * X, 'last': bool, 'got': bool} while BackwardProjection.forward runs its transformer -- the LAST
#                           layer's tail + FFN kernel may then write the refined BEV as (B, C, Y, X) planes itself (fbbev_rows_tail_ffn_x3_planes)
OUT_PLANES = _os0.environ.get('FBBEV_BP_OUT_PLANES', '1') != '0'    # A/B knob
_NORMED = object()       # 'residual' slot of a deferred branch result whose following LayerNorm already ran inside the branch's last GEMM


def host_values(t):
    """The Python values a const_tensor was built from (None for any other tensor)."""
    return getattr(t, '_fbbev_host', None)


def inv3x3(m):
    """Closed-form inverse of (...,3,3) matrices (adjugate / determinant): no solver dependency."""
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    A, Bc, C = e * i - f * h, -(d * i - f * g), d * h - e * g
    det = a * A + b * Bc + c * C
    adj = torch.stack([torch.stack([A, -(b * i - c * h), b * f - c * e], -1),
                       torch.stack([Bc, a * i - c * g, -(a * f - c * d)], -1),
                       torch.stack([C, -(a * h - b * g), a * e - b * d], -1)], -2)
    return adj / det[..., None, None]


def _xavier(m, bias=0.):
    if m is not None:
        nn.init.xavier_uniform_(m.weight)
        nn.init.constant_(m.bias, bias)


@register
class CustormLearnedPositionalEncoding(nn.Module):
    """positional_encoding.py:11-68."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)
        self.num_feats = num_feats

    def _table(self, h, w, device):
        x_embed = self.col_embed(torch.arange(w, device=device))
        y_embed = self.row_embed(torch.arange(h, device=device))
        return torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)), dim=-1)

    def forward(self, bs, h, w, device):
        rw, cw = self.row_embed.weight, self.col_embed.weight
        if torch.is_grad_enabled() and (rw.requires_grad or cw.requires_grad):
            pos = self._table(h, w, device)
        else:
            # inference: the (h, w, C) table is a per-shape constant of the two embedding tables -- built once per weight version
            # (data_ptr + _version, as the other folded-weight caches) instead of seven ATen launches per step
            key = (h, w, str(device), rw.data_ptr(), rw._version, cw.data_ptr(), cw._version)
            if getattr(self, '_pos_key', None) != key:
                with torch.no_grad():
                    self._pos = self._table(h, w, device).contiguous()
                self._pos_key = key
            pos = self._pos
        # (bs,C,h,w) like the reference (positional_encoding.py:57-60) but as a broadcast VIEW of the token-major
        # (h,w,C) table instead of `.repeat`: flattened and permuted back to (bs,Q,C) by the encoder it is again
        # contiguous rows, so `query + query_pos` runs vectorised and nothing is copied
        return pos.permute(2, 0, 1).unsqueeze(0).expand(bs, -1, -1, -1)


@register
class FFN(nn.Module):
    """mmcv.cnn.bricks.transformer.FFN (state_dict: layers.0.0.*, layers.1.*)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=None, ffn_drop=0.,
                 dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        layers, in_ch = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(Linear(in_ch, feedforward_channels), nn.ReLU(inplace=True),
                                        nn.Dropout(ffn_drop)))
            in_ch = feedforward_channels
        layers.append(Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity
        self.embed_dims = embed_dims

    def _has_live_dropout(self):
        """some Dropout of this FFN has p > 0 (it acts whenever the module is in training mode, grad or no grad)"""
        return any(isinstance(l, nn.Dropout) and l.p > 0 for l in self.layers.modules())

    def _one_kernel(self, x, identity, defer, norm, real):
        """inference, the encoder's shape (Linear + ReLU, Linear; in <= 96, hidden % 64 == 0, out <= 80): both GEMMs, the ReLU, the
        residual and -- when the layer hands it over -- the following LayerNorm in ONE kernel (fbbev_rows_ffn_x3): the hidden rows stay
        in LDS.  None when the shape / route does not apply."""
        if not (FUSE_FFN and len(real) == 2 and isinstance(real[0], nn.Sequential) and isinstance(real[0][0], Linear)
                and isinstance(real[0][1], nn.ReLU) and isinstance(real[1], Linear)):
            return None
        l1, l2 = real[0][0], real[1]
        I, H, O = l1.in_features, l1.out_features, l2.out_features
        if not (x3_ok(x, I, H) and I <= 96 and H % 64 == 0 and O <= 80 and O % 4 == 0 and l1.bias is not None and l2.bias is not None
                and x.is_contiguous() and x.data_ptr() % 16 == 0):     # a misaligned view takes the two-GEMM fallback
            return None
        res = None
        if self.add_identity:
            res = (x if identity is None else identity)
            if res.dtype != torch.float32 or res.shape[:-1] != x.shape[:-1] or res.shape[-1] != O:
                return None
            res = res.contiguous().reshape(-1, O)
            if res.data_ptr() % 16 != 0:
                return None
        fuse_norm = defer and norm is not None and self.add_identity and _RL.ln_fusable(norm, None, x, O)
        if defer and not fuse_norm and self.add_identity:
            res_arg, tail = None, (x if identity is None else identity)      # the caller's LayerNorm adds the residual itself
        else:
            res_arg, tail = res, None
        for lin in (l1, l2):
            if not hasattr(lin, '_x3'):
                lin._x3 = X3Weights()
        c1, c2 = l1._x3.get(l1.weight, l1.bias), l2._x3.get(l2.weight, l2.bias)
        y = _capi.rows_ffn_x3(x.reshape(-1, I), c1.frag, c1.b, c2.frag, c2.b, H, O, residual=res_arg,
                              ln_weight=norm.weight if fuse_norm else None, ln_bias=norm.bias if fuse_norm else None,
                              eps=norm.eps if fuse_norm else 1e-5).view(*x.shape[:-1], O)
        if fuse_norm:
            return y, _NORMED
        if defer:
            return y, tail
        return y

    def fused_tail_spec(self, E):
        """(W1 fragments, W2 fragments, hidden width) when this FFN is the shape fbbev_rows_tail_ffn_x3 runs behind an attention
        block's tail -- Linear(E, H) + ReLU, Linear(H, E), add_identity, H % 64 == 0, E % 16 == 0, E <= 80 -- else None."""
        if self.training and self._has_live_dropout():
            # train() + no_grad() with ffn_drop > 0 must keep the Dropout layers (FFN.forward keeps `self.layers` then)
            return None
        real = [l for l in self.layers if not isinstance(l, nn.Dropout)]
        if not (FUSE_TAIL_FFN and FUSE_FFN and self.add_identity and len(real) == 2 and isinstance(real[0], nn.Sequential)
                and isinstance(real[0][0], Linear) and isinstance(real[0][1], nn.ReLU) and isinstance(real[1], Linear)):
            return None
        l1, l2 = real[0][0], real[1]
        if (l1.in_features != E or l2.out_features != E or l2.in_features != l1.out_features or l1.out_features % 64 != 0 or
                E % 16 != 0 or E > 80 or l1.bias is None or l2.bias is None or not l1.weight.is_cuda):
            return None
        for lin in (l1, l2):
            if not hasattr(lin, '_x3'):
                lin._x3 = X3Weights()
        return l1._x3.get(l1.weight, l1.bias), l2._x3.get(l2.weight, l2.bias), l1.out_features

    def forward(self, x, identity=None, _defer_residual=False, _norm=None):
        if torch.is_grad_enabled() or self.training:
            out = self.layers(x)
        else:
            real = [l for l in self.layers if not isinstance(l, nn.Dropout)]
            one = self._one_kernel(x, identity, _defer_residual, _norm, real)
            if one is not None:
                return one
            # inference: Linear + ReLU blocks as one call each (the ReLU rides in the GEMM's store epilogue; dropout is identity)
            out = x
            for layer in real:
                if isinstance(layer, nn.Sequential) and isinstance(layer[0], Linear) and isinstance(layer[1], nn.ReLU):
                    out = layer[0](out, relu=True)
                elif (layer is real[-1] and _norm is not None and _defer_residual and self.add_identity and isinstance(layer, Linear)):
                    # the last Linear + residual + the layer's following LayerNorm in one kernel (fbbev_rows_linear_x3_ln)
                    res = x if identity is None else identity
                    return layer(out, ln=(res.contiguous(), _norm)), _NORMED
                else:
                    out = layer(out)
        if not self.add_identity:
            return (out, None) if _defer_residual else out
        res = x if identity is None else identity
        # _defer_residual (BEVFormerEncoderLayer's inference route): hand (branch output, residual) to the following LayerNorm,
        # whose kernel adds them while it reads the row -- the separate element-wise add pass is skipped
        return (out, res) if _defer_residual else res + out


def _ring_offsets(num_heads):
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    g = torch.stack([thetas.cos(), thetas.sin()], -1)
    return g / g.abs().max(-1, keepdim=True)[0]


@register
class MultiScaleDeformableAttention(nn.Module):
    """mmcv.ops.MultiScaleDeformableAttention (BEV self-attention of the encoder layer, cfg :176-180)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.im2col_step, self.batch_first = im2col_step, batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.output_proj = Linear(embed_dims, embed_dims)
        self.fused_inference = True
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        g = _ring_offsets(self.num_heads).view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            g[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = g.view(-1)
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        _xavier(self.value_proj)
        _xavier(self.output_proj)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, _defer_residual=False, _norm=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        pos = None
        if query_pos is not None:
            # inference (batch_first rows): the two projections add the positional rows while they load the query rows
            if (self.batch_first and self.fused_inference and not torch.is_grad_enabled() and key_padding_mask is None and
                    x3_ok(query, query.shape[-1], 4)):
                pos = query_pos
            else:
                query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        num_value = value.shape[1]
        Dh = self.embed_dims // self.num_heads
        needs_grad = torch.is_grad_enabled() and (query.requires_grad or value.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if (self.fused_inference and not needs_grad and query.is_cuda and key_padding_mask is None
                and Dh in (4, 8, 10, 16, 32) and query.dtype == torch.float32
                and not (self.training and self.dropout.p > 0) and reference_points.shape[-1] == 2):
            # inference: fbbev_msda_fwd_fused builds `reference_points + offsets / (W,H)` per sample inside the kernel
            # (the reference spends two elementwise passes on that tensor) and reads head-padded value rows with
            # aligned 16-byte loads (value_proj rows padded once, like DA_MSDeformableAttention)
            hw = host_values(spatial_shapes)
            if (_RL.X3 and self.batch_first and hw is not None and len(hw) == 1 and num_query == hw[0][0] * hw[0][1]
                    and num_value == num_query and hw[0][1] >= 2
                    and tuple(reference_points.shape[1:]) == (num_query, 1, 2) and self.embed_dims % 8 == 0
                    and _capi.msda_self_fused_supported(bs, num_value, self.num_heads, Dh, self.num_levels, num_query,
                                                        self.num_points, hw[0][1])):
                # round 4: query rows -> attention output in ONE kernel (fbbev_msda_self_fused): value_proj writes head planes,
                # the sampling_offsets / attention_weights projections and the softmax run inside the sampler's workgroups
                if not hasattr(self, '_vx3p'):
                    self._vx3p, self._so_x3p, self._aw_x3p = X3Weights(), X3Weights(), X3Weights()
                same = value is query
                if not query.is_contiguous():             # the encoder's first layer hands a (Q, B, C) tensor permuted to (B, Q, C)
                    query = query.contiguous()
                value = query if same else value.contiguous()
                vp = self._vx3p.get(self.value_proj.weight, self.value_proj.bias)
                planes = _capi.rows_linear_x3_planes(value.reshape(bs * num_value, self.embed_dims), vp.frag, vp.b, num_value,
                                                     self.num_heads, Dh)
                so_c = self._so_x3p.get(self.sampling_offsets.weight, self.sampling_offsets.bias)
                aw_c = self._aw_x3p.get(self.attention_weights.weight, self.attention_weights.bias)
                q_in, add = query, None
                if pos is not None:
                    add = _RL._addend_rows(pos, query) if _RL.FOLD_ADDEND else None
                    if add is None:
                        q_in = query + pos
                out = torch.empty(bs, num_query, self.embed_dims, dtype=torch.float32, device=query.device)
                rk = (reference_points.data_ptr(), reference_points._version, tuple(reference_points.shape), bs, str(query.device))
                if getattr(self, '_ref_key', None) != rk:      # the encoder's cached 2-D grid: one contiguous copy per shape
                    self._ref, self._ref_src, self._ref_key = reference_points.expand(bs, num_query, 1, 2).contiguous(), reference_points, rk
                ref = self._ref
                if (_norm is not None and _defer_residual and FUSE_ATTN_TAIL and self.embed_dims % 16 == 0
                        and _RL.ln_fusable(_norm, identity.contiguous(), out, self.embed_dims)):
                    # output_proj + residual + the following LayerNorm inside the attention kernel's workgroups (fbbev_msda_self_fused_ln):
                    # the attention output never leaves the CU
                    if not hasattr(self.output_proj, '_x3'):
                        self.output_proj._x3 = X3Weights()
                    oc = self.output_proj._x3.get(self.output_proj.weight, self.output_proj.bias)
                    _capi.msda_self_fused(planes, ref, q_in, add, so_c.frag, so_c.b, aw_c.frag, aw_c.b, self.num_points, hw[0][1],
                                          hw[0], out, out_proj=(oc.frag, oc.b, identity.contiguous(), _norm.weight, _norm.bias, _norm.eps))
                    return out, _NORMED
                _capi.msda_self_fused(planes, ref, q_in, add, so_c.frag, so_c.b, aw_c.frag, aw_c.b, self.num_points, hw[0][1],
                                      hw[0], out)
                if _norm is not None and _defer_residual:     # output_proj + residual + the following LayerNorm: one kernel
                    return self.output_proj(out, ln=(identity.contiguous(), _norm)), _NORMED
                out = self.output_proj(out)
                return (out, identity) if _defer_residual else out + identity
            HS = (Dh + 3) // 4 * 4
            w, b = self.value_proj.weight, self.value_proj.bias
            # data_ptr: a storage swap (param.data = ..., load_state_dict(assign=True), EMA) does not bump _version
            key = (w._version, b._version, w.data_ptr(), b.data_ptr(), w.device)
            if getattr(self, '_vpad_key', None) != key:
                with torch.no_grad():
                    self._vpad = _pad_interleave_rows(w, b, self.num_heads, Dh, HS)
                self._vpad_key = key
            w, b = self._vpad
            if x3_ok(value, w.shape[1], w.shape[0]):
                if not hasattr(self, '_vx3'):
                    self._vx3 = X3Weights()
                M_ = self.num_heads
                value = linear_x3(value, self._vx3.get(self.value_proj.weight, self.value_proj.bias,
                                                      lambda w_, b_: _pad_interleave_rows(w_, b_, M_, Dh, HS)))
                value = value.view(bs, num_value, self.num_heads, HS)
            else:
                value = F.linear(value, w, b).view(bs, num_value, self.num_heads, HS)   # stored (HS/4, M, 4) per token
            so = self.sampling_offsets(query, addend=pos)
            aw = self.attention_weights(query, addend=pos).view(bs, num_query, self.num_heads, -1).softmax(-1)
            aw = aw.view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
            out = torch.empty(bs, num_query, self.embed_dims, dtype=torch.float32, device=query.device)
            ref = reference_points.expand(bs, num_query, self.num_levels, 2).contiguous()
            _capi.msda_fwd_fused(value, spatial_shapes.to(torch.int64).contiguous(),
                                 level_start_index.to(torch.int64).contiguous(), ref, so.contiguous(), aw.contiguous(),
                                 out, head_dim=Dh, value_interleaved=True)
            out = self.output_proj(out)
            if not self.batch_first:
                out = out.permute(1, 0, 2)
            return (out, identity) if _defer_residual else out + identity
        if pos is not None:                                  # the fused branch was not taken after all
            query = query + pos
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        so = self.sampling_offsets(query).view(bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        aw = self.attention_weights(query).view(bs, num_query, self.num_heads, self.num_levels * self.num_points)
        aw = aw.softmax(-1).view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
        assert reference_points.shape[-1] == 2
        norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + so / norm[None, None, None, :, None, :]
        out = MultiScaleDeformableAttnFunction_fp32.apply(value, spatial_shapes, level_start_index, loc, aw,
                                                          self.im2col_step)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return (self.dropout(out), identity) if _defer_residual else self.dropout(out) + identity


@register
class DA_MSDeformableAttention(nn.Module):
    """spatial_cross_attention_depth.py:361-601 (no output_proj, :406)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, num_Z_anchors=4, im2col_step=64,
                 dropout=0.1, batch_first=True, disable_deformable=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.num_Z_anchors, self.im2col_step, self.batch_first = num_Z_anchors, im2col_step, batch_first
        self.disable_deformable = disable_deformable
        self.output_proj = None
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        """:440-462 -- ring pattern, scaled by (point index within anchor + 1)."""
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        self.each_anchor_points = self.num_points // self.num_Z_anchors
        g = _ring_offsets(self.num_heads).view(self.num_heads, 1, 1, 1, 2).repeat(
            1, self.num_levels, self.each_anchor_points, self.num_Z_anchors, 1)
        for i in range(self.each_anchor_points):
            g[:, :, i, :, :] *= i + 1
        self.sampling_offsets.bias.data = g.view(-1)
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        _xavier(self.value_proj)

    def project(self, query):
        """Per-query projections (camera independent): raw offsets (B,Q,M,L,P,2), softmaxed weights."""
        bs, nq, _ = query.shape
        so = self.sampling_offsets(query).view(bs, nq, self.num_heads, self.num_levels, self.num_points, 2)
        aw = self.attention_weights(query).view(bs, nq, self.num_heads, self.num_levels * self.num_points)
        if self.disable_deformable:
            so, aw = so * 0, aw * 0
        aw = aw.softmax(-1).view(bs, nq, self.num_heads, self.num_levels, self.num_points)
        return so, aw

    def project_head_minor(self, query, softmax=True, addend=None):
        """`project` with the sampling_offsets rows permuted so that the offsets come out head-minor, (B,Q,L,P,M,2):
        the layout the fused kernel reads with contiguous lanes (same dot product per element, only the row order of
        the weight matrix changes).  The attention weights keep (B,Q,M,L,P): their softmax runs over the last dim."""
        bs, nq, _ = query.shape
        M, L, P = self.num_heads, self.num_levels, self.num_points
        key = (M, L, P, str(query.device))
        if getattr(self, '_perm_key', None) != key:
            o = torch.arange(M * L * P).view(M, L, P).permute(1, 2, 0).reshape(-1)        # new (l,p,m) -> old (m,l,p)
            self._perm_so = (o[:, None] * 2 + torch.arange(2)[None]).reshape(-1).to(query.device)
            self._perm_key = key
        if not hasattr(self, '_so_x3'):
            self._so_x3 = X3Weights()
        perm = self._perm_so
        # addend: the positional encoding the caller would otherwise have added to `query` in a pass of its own
        so = linear_rows(query, self.sampling_offsets.weight, self.sampling_offsets.bias, cache=self._so_x3,
                         transform=lambda w_, b_: (w_[perm], b_[perm]), addend=addend)
        so = so.view(bs, nq, L, P, M, 2)
        aw = self.attention_weights(query, addend=addend).view(bs, nq, M, L * P)
        if self.disable_deformable:
            so, aw = so * 0, aw * 0
        # softmax=False: raw logits for a kernel that applies the softmax itself (fbbev_da_cross_attn_fwd_zt)
        return so, (aw.softmax(-1) if softmax else aw).view(bs, nq, M, L, P)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, bev_query_depth=None,
                pred_img_depth=None, **kwargs):
        """Composite (autograd) path == the reference's CUDA branch (:513-595)."""
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        num_value = value.shape[1]
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        so, aw = self.project(query)
        norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        Za = reference_points.shape[2]
        P = self.num_points
        so = (so / norm[None, None, None, :, None, :]).view(bs, nq, self.num_heads, self.num_levels, P // Za, Za, 2)
        loc = (reference_points[:, :, None, None, None, :, :] + so).view(bs, nq, self.num_heads, self.num_levels, P, 2)
        Fn = MultiScaleDeformableAttnFunction_fp32
        dref = reference_points.reshape(bs, nq * Za, 1, 1, 1, 2).contiguous()
        dsamp = Fn.apply(pred_img_depth.unsqueeze(2).contiguous(), spatial_shapes[0:1], level_start_index[0:1], dref,
                         torch.ones_like(dref[..., 0]).contiguous(), self.im2col_step).reshape(bs, nq, Za, -1)
        dw = (dsamp * bev_query_depth).sum(-1)
        dw = dw.unsqueeze(2).repeat(1, 1, P // Za, 1).reshape(bs, nq, P)
        aw = aw * dw[:, :, None, None, :]
        out = Fn.apply(value, spatial_shapes, level_start_index, loc.contiguous(), aw.contiguous(), self.im2col_step)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out


class FusedDACrossAttention(torch.autograd.Function):
    """fbbev_da_cross_attn_fwd / fbbev_da_cross_attn_bwd as one differentiable op: gradients for the projected
    camera tokens (`value`), the depth distribution, the raw sampling offsets and the softmaxed attention weights; the
    geometry inputs (reference points, masks, query depths) carry none, as in the reference (they come from
    point_sampling on camera parameters).  No host sync in either direction."""

    @staticmethod
    def forward(ctx, value, pred_depth, offsets, attn, spatial_shapes, level_start_index, ref_cam, mask, qdepth,
                d0, dstep, head_minor, head_dim=None, level_hw=None, bev_w=0):
        B, Q = offsets.shape[0], offsets.shape[1]
        M = value.shape[2]
        Dh = value.shape[3] if head_dim is None else head_dim          # value rows may be head-padded (stride value.shape[3])
        slots = torch.empty((B, Q, M * Dh), dtype=torch.float32, device=value.device)
        Ncam, Za = mask.shape[0], mask.shape[3]
        L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
        if (value.is_cuda and level_hw is not None and min(int(w) for _, w in level_hw) >= 2 and
                _capi.da_cross_attn_fwd_planes_supported(B, Ncam, value.shape[1], M, Dh, L, Q, P, Za) and
                # the entry wants 16-byte aligned records / 4-byte aligned mask words; a storage-offset view takes the row kernel
                ref_cam.data_ptr() % 16 == 0 and qdepth.data_ptr() % 16 == 0 and mask.data_ptr() % 4 == 0):
            # round 4: the sampler's mapping for the training forward too -- tokens as head planes, a wave = one head of an
            # 8 x 8 patch of queries (k_da_fwd_planes); the row kernel stays for every other shape
            planes = _capi.value_rows_to_head_planes(value, head_dim=Dh, interleaved=bool(head_minor & 4))
            _capi.da_cross_attn_fwd_planes(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn,
                                           d0, dstep, slots, head_minor=head_minor, bev_w=bev_w,
                                           min_level_width=min(int(w) for _, w in level_hw))
        else:
            _capi.da_cross_attn_fwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                                    attn, d0, dstep, slots, head_minor=head_minor, head_dim=Dh)
        ctx.save_for_backward(value, pred_depth, offsets, attn, spatial_shapes, level_start_index, ref_cam, mask, qdepth)
        ctx.consts = (d0, dstep, head_minor, Dh, level_hw, int(bev_w or 0))
        return slots

    @staticmethod
    def backward(ctx, grad_slots):
        value, pred_depth, offsets, attn, spatial_shapes, level_start_index, ref_cam, mask, qdepth = ctx.saved_tensors
        d0, dstep, head_minor, Dh, level_hw, bev_w = ctx.consts
        gv, gd, go, ga = (torch.zeros_like(t) for t in (value, pred_depth, offsets, attn))
        _capi.da_cross_attn_bwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                                attn, grad_slots.contiguous().float(), d0, dstep, head_minor, gv, gd, go, ga, head_dim=Dh,
                                level_hw=level_hw, bev_w=bev_w)       # the BEV row length: the unit gradients take 8 x 8 query patches
        return gv, gd, go, ga, None, None, None, None, None, None, None, None, None, None, None


def _pad_interleave_rows(w, b, M, Dh, HS, interleave=True, piece=4):
    """value_proj rows for the fused kernels: each head padded Dh -> HS (multiple of 4) output rows, then the rows of a
    token reordered (head m, chunk k, e) -> (chunk k, head m, e): the 8 head lanes of a query read one contiguous
    M*16-byte piece per load instruction (k_da_cross_attn_fwd_unit, QI).  Same dot products, only the row order of the
    weight matrix changes; differentiable (pad + permute of the parameter)."""
    E = w.shape[1]
    w, b = F.pad(w.view(M, Dh, E), (0, 0, 0, HS - Dh)), F.pad(b.view(M, Dh), (0, HS - Dh))
    if interleave:                       # piece = elements of a (chunk, head) piece: 4 floats or 8 16-bit elements = 16 bytes
        w, b = w.view(M, HS // piece, piece, E).permute(1, 0, 2, 3), b.view(M, HS // piece, piece).permute(1, 0, 2)
    return w.reshape(M * HS, E).contiguous(), b.reshape(M * HS).contiguous()


import os as _os
FUSE_FFN = _os.environ.get('FBBEV_FUSE_FFN', '1') != '0'               # the FFN pair as one kernel (fbbev_rows_ffn_x3; A/B knob)
FUSE_TAIL_FFN = _os.environ.get('FBBEV_FUSE_TAIL_FFN', '1') != '0'   # cross-attention tail + FFN block as one kernel (fbbev_rows_tail_ffn_x3; A/B knob)
PLANES_16BIT = _os.environ.get('FBBEV_DA_16BIT_PLANES', '1') != '0'   # 16-bit camera tokens as head planes on the one-kernel sampler (A/B knob; 0: round-3 kernels)
FUSE_OUT_NORM = _os.environ.get('FBBEV_FUSE_OUT_NORM', '1') != '0'     # output_proj / FFN tail + residual + LayerNorm in one kernel (A/B knob)
FUSE_ATTN_TAIL = _os.environ.get('FBBEV_FUSE_ATTN_TAIL', '1') != '0'   # ... inside the attention kernel's own workgroups (A/B knob)
FUSE_ATTN_TAIL_DA = _os.environ.get('FBBEV_FUSE_ATTN_TAIL_DA', '0') != '0'   # the same for the cross-attention (needs its 8-heads-per-workgroup form)
FUSED_BWD_MAX_HEAD_DIM = 32                 # k_da_cross_attn_bwd: one lane per channel, groups of 16 / 32 lanes


@register
class DA_SpatialCrossAttention(nn.Module):
    """spatial_cross_attention_depth.py:31-223."""

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None, batch_first=False,
                 deformable_attention=dict(type='DA_MSDeformableAttention', embed_dims=256, num_levels=4),
                 layer_scale=None, dbound=None, fused=True, **kwargs):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.deformable_attention = build(deformable_attention)
        self.embed_dims, self.num_cams, self.dbound, self.batch_first = embed_dims, num_cams, dbound, batch_first
        self.output_proj = Linear(embed_dims, embed_dims)
        self.layer_scale = nn.Parameter(layer_scale * torch.ones(embed_dims)) if layer_scale is not None else None
        self.fused = fused
        # inference option (not part of the reference config): camera tokens (value_proj output) kept in bf16 / fp16 for the
        # fused sampling kernel -- half the gather bytes, fp32 accumulate; None = fp32 (the reference's precision)
        self.value_dtype = None
        _xavier(self.output_proj)

    # ---- inference: one fused HIP launch (fbbev_da_cross_attn_fwd), no host sync
    def _slots_fused(self, query, value, reference_points_cam, mask, bev_query_depth, pred_img_depth,
                     spatial_shapes, level_start_index, bev_w=0, query_pos=None, _tail=None):
        da = self.deformable_attention
        B, Q, E = query.shape
        ncam, S, _, _ = value.shape
        M = da.num_heads
        Dh = E // M
        HS = (Dh + 3) // 4 * 4
        x = value.permute(2, 0, 1, 3).reshape(B * ncam, S, E)
        # value_proj with its output rows padded per head (Dh = 10 -> 12 floats) and stored chunk-major per token
        # (_pad_interleave_rows): every 4-channel chunk of a head is 16-byte aligned and the 8 heads' chunks are adjacent.
        # Same dot products for the real rows; the padding rows are zero and ignored by the kernel.
        wt, bs = da.value_proj.weight, da.value_proj.bias
        interleave = True
        # every parameter of the deformable attention counts: with only sampling_offsets / attention_weights
        # trainable the zero-token inference branch below would hand back slots without an autograd node
        grad_mode = torch.is_grad_enabled() and (value.requires_grad or query.requires_grad or pred_img_depth.requires_grad or
                                                 (query_pos is not None and query_pos.requires_grad) or
                                                 any(p.requires_grad for p in da.parameters()))
        if (PLANES_16BIT and self.value_dtype in (torch.bfloat16, torch.float16) and not grad_mode and not da.disable_deformable and
                x.dtype == torch.float32 and x.is_cuda):
            # round 5: 16-bit camera tokens on the one-kernel route too -- value_proj writes bf16 / fp16 HEAD PLANES
            # (fbbev_rows_linear_x3_planes_e), the sampler reads them (fbbev_da_cross_attn_fused_e): half the gather bytes of the
            # default, fp32 products and sums.  None (unsupported shape): the round-3 kernels below
            slots = self._slots_one_kernel(da, x, query, query_pos, reference_points_cam, mask, bev_query_depth, pred_img_depth,
                                           spatial_shapes, level_start_index, bev_w, plane_dtype=self.value_dtype)
            if slots is not None:
                return slots
        if self.value_dtype in (torch.bfloat16, torch.float16) and not grad_mode and Dh in (8, 10, 16, 32):
            # 16-bit tokens: rows chunk-major with 8-element pieces, rounded once after the fp32 projection
            HS16 = (Dh + 7) // 8 * 8
            key = (wt.data_ptr(), wt._version, bs._version, str(wt.device), self.value_dtype)
            if getattr(self, '_vpad16_key', None) != key:
                with torch.no_grad():
                    self._vpad16 = _pad_interleave_rows(wt, bs, M, Dh, HS16, True, piece=8)
                self._vpad16_key = key
            w, bb = self._vpad16
            v = F.linear(x, w, bb).to(self.value_dtype).view(B * ncam, S, M, HS16)
            so, aw = da.project_head_minor(query, addend=query_pos)
            DC, H0, W0 = pred_img_depth.shape[2:]
            slots = torch.empty((B, Q, M * Dh), dtype=torch.float32, device=query.device)
            _capi.da_cross_attn_fwd(v, spatial_shapes.to(torch.int64).contiguous(), level_start_index.to(torch.int64).contiguous(),
                                    pred_img_depth.reshape(B * ncam, DC, H0, W0).contiguous().float(),
                                    reference_points_cam.contiguous().float(), mask.contiguous(),
                                    bev_query_depth.squeeze(-1).contiguous().float(), so.contiguous().float(),
                                    aw.contiguous().float(), self.dbound[0], self.dbound[2], slots, head_minor=1 | 4, head_dim=Dh)
            return slots
        if torch.is_grad_enabled() and (wt.requires_grad or bs.requires_grad or value.requires_grad):
            # training: the backward keeps the value gradient in LDS planes when a head's plane fits (single-level FB-OCC
            # shapes); otherwise it uses global atomics, whose 10 consecutive channel lanes want head-major rows
            interleave = _capi.da_cross_attn_bwd_ws_bytes(B, ncam, S, M, Dh, Q, HS, da.num_levels, da.num_points,
                                                          host_values(spatial_shapes)) > 0
            w, bb = _pad_interleave_rows(wt, bs, M, Dh, HS, interleave)
        else:                               # inference: once per weight version
            key = (wt.data_ptr(), wt._version, bs._version, str(wt.device))
            if getattr(self, '_vpad_key', None) != key:
                with torch.no_grad():
                    self._vpad = _pad_interleave_rows(wt, bs, M, Dh, HS)
                self._vpad_key = key
            w, bb = self._vpad
        DC, H0, W0 = pred_img_depth.shape[2:]
        zt = not grad_mode and x.dtype == torch.float32 and x.is_cuda
        hm = 1 | (4 if interleave else 0)
        # the pipelined kernel applies the attention softmax while it stages the weights: hand it the raw logits
        fuse_sm = zt and not da.disable_deformable and _capi.da_fuses_softmax(
            B, ncam, S, M, Dh, da.num_levels, Q, da.num_points, reference_points_cam.shape[3], hm, HS)
        if zt and not da.disable_deformable:
            slots = self._slots_one_kernel(da, x, query, query_pos, reference_points_cam, mask, bev_query_depth, pred_img_depth,
                                           spatial_shapes, level_start_index, bev_w, _tail=_tail)
            if slots is not None:
                return slots
        so, aw = da.project_head_minor(query, softmax=not fuse_sm, addend=query_pos)
        if zt:
            # inference: the projection writes into a buffer with one extra all-zero token behind the rows -- the pipelined
            # sampler (fbbev_da_cross_attn_fwd_zt: two samples in flight per lane) reads it for padded corners and
            # out-of-image samples instead of branching around their loads
            _, rows = _capi.da_value_buffer(B * ncam * S, M * HS, x.device)
            if x3_ok(x, E, M * HS):
                if not hasattr(self, '_vx3'):
                    self._vx3 = X3Weights()
                linear_x3(x.reshape(B * ncam * S, E),
                          self._vx3.get(wt, bs, lambda w_, b_: _pad_interleave_rows(w_, b_, M, Dh, HS)), out=rows)
            else:
                torch.addmm(bb, x.reshape(B * ncam * S, E), w.t(), out=rows)
            slots = torch.empty((B, Q, M * Dh), dtype=torch.float32, device=query.device)
            _capi.da_cross_attn_fwd(rows.view(B * ncam, S, M, HS), spatial_shapes.to(torch.int64).contiguous(),
                                    level_start_index.to(torch.int64).contiguous(),
                                    pred_img_depth.reshape(B * ncam, DC, H0, W0).contiguous().float(),
                                    reference_points_cam.contiguous().float(), mask.contiguous(),
                                    bev_query_depth.squeeze(-1).contiguous().float(), so.contiguous().float(),
                                    aw.contiguous().float(), self.dbound[0], self.dbound[2], slots,
                                    head_minor=hm | (_capi.DA_ATTN_LOGITS if fuse_sm else 0), head_dim=Dh, zero_token=True,
                                    bev_w=bev_w)
            return slots
        v = linear_rows(x, w, bb).view(B * ncam, S, M, HS)        # a token's M*HS floats are (HS/4, M, 4)
        return FusedDACrossAttention.apply(
            v.contiguous().float(), pred_img_depth.reshape(B * ncam, DC, H0, W0).contiguous().float(),
            so.contiguous().float(), aw.contiguous().float(), spatial_shapes.to(torch.int64).contiguous(),
            level_start_index.to(torch.int64).contiguous(), reference_points_cam.contiguous().float(), mask.contiguous(),
            bev_query_depth.squeeze(-1).contiguous().float(), self.dbound[0], self.dbound[2], 1 | (4 if interleave else 0), Dh,
            host_values(spatial_shapes), bev_w or 0)

    # ---- inference default since round 4: query rows -> slots in ONE kernel (fbbev_da_cross_attn_fused, da_fused_kernels.h)
    def _slots_one_kernel(self, da, x, query, query_pos, reference_points_cam, mask, bev_query_depth, pred_img_depth,
                          spatial_shapes, level_start_index, bev_w, _tail=None, plane_dtype=None):
        """value_proj writes the camera tokens as head planes (fbbev_rows_linear_x3_planes); the sampling_offsets / attention_weights
        projections, the softmax and the sampling run inside one kernel from the query rows (+ positional rows): no offsets /
        weights tensors (492 MB written and re-read at BASELINE configs[2], B = 4).  None when the shape is not the kernel's
        (M = 8, Dh in {8, 10}, 8 points, 4 anchors, levels >= 2 tokens wide, a BEV grid) or FBBEV_ROWS_LINEAR=f32 asks for the
        vendor fp32 GEMMs: the caller falls through to the projection kernels + the pipelined sampler."""
        B, Q, E = query.shape
        BN, S, _ = x.shape
        ncam = BN // B
        M, L, P = da.num_heads, da.num_levels, da.num_points
        Dh = E // M
        hw = host_values(spatial_shapes)
        if (not _RL.X3 or hw is None or min(int(w) for _, w in hw) < 2 or E % 8 != 0 or not query.is_contiguous() or
                not _capi.da_cross_attn_fused_supported(B, ncam, S, M, Dh, L, Q, P, reference_points_cam.shape[3], bev_w or 0)):
            return None
        if not hasattr(self, '_vx3p'):
            self._vx3p, da._so_x3p, da._aw_x3p = X3Weights(), X3Weights(), X3Weights()
        pre = _PRE.planes.get(id(self)) if (_PRE is not None and plane_dtype is None) else None
        if pre is not None and pre[0] == (x.data_ptr(), BN, S, E):      # projected on the side stream by BackwardProjection.prefetch
            planes = pre[1]
        else:
            planes = self._value_planes(x.reshape(BN * S, E), S, plane_dtype)
        if plane_dtype is not None:
            _tail = None                                                  # (the block tail inside the sampler is an fp32-plane route)
        so = da._so_x3p.get(da.sampling_offsets.weight, da.sampling_offsets.bias)
        aw = da._aw_x3p.get(da.attention_weights.weight, da.attention_weights.bias)
        addend = None
        if query_pos is not None:
            addend = _RL._addend_rows(query_pos, query) if _RL.FOLD_ADDEND else None
            if addend is None:
                query = query + query_pos
        DC, H0, W0 = pred_img_depth.shape[2:]
        slots = torch.empty((B, Q, E), dtype=torch.float32, device=query.device)
        return _capi.da_cross_attn_fused(
            planes, spatial_shapes.to(torch.int64).contiguous(), level_start_index.to(torch.int64).contiguous(),
            pred_img_depth.reshape(BN, DC, H0, W0).contiguous().float(), reference_points_cam.contiguous().float(), mask.contiguous(),
            bev_query_depth.squeeze(-1).contiguous().float(), query, addend, so.frag, so.b, aw.frag, aw.b, P, self.dbound[0],
            self.dbound[2], bev_w, min(int(w) for _, w in hw), slots, out_proj=self._take_tail(_tail))

    def _value_planes(self, x2d, S, dtype=None):
        """value_proj of the camera-token rows (images * S, E) written as head planes (images, M, S, Dh); dtype bf16 / fp16: stored
        in 16 bits (the fp32 projection rounded once)"""
        da = self.deformable_attention
        if not hasattr(self, '_vx3p'):
            self._vx3p, da._so_x3p, da._aw_x3p = X3Weights(), X3Weights(), X3Weights()
        vp = self._vx3p.get(da.value_proj.weight, da.value_proj.bias)
        return _capi.rows_linear_x3_planes(x2d, vp.frag, vp.b, S, da.num_heads, self.embed_dims // da.num_heads, dtype=dtype)

    @staticmethod
    def _take_tail(tail):
        if tail is None:
            return None
        tail['done'] = True
        return tail['spec']

    # ---- training: vectorised rebatch + composite deformable attention (autograd through the MSDA op)
    def _slots_composite(self, query, value, reference_points_cam, mask, bev_query_depth, pred_img_depth,
                         spatial_shapes, level_start_index, bev_w=0):
        B, Q, E = query.shape
        ncam, S, _, _ = value.shape
        DC = pred_img_depth.shape[2]
        hit = mask.any(-1).permute(1, 0, 2)                                    # (B,N,Q)
        lens = hit.sum(-1)
        max_len = max(int(lens.max()), 1)                                      # the ONE host sync
        order = torch.argsort((~hit).to(torch.uint8), dim=-1, stable=True)[..., :max_len]   # hit queries first, ascending
        valid = torch.arange(max_len, device=query.device)[None, None] < lens[..., None]
        gi = order[..., None]
        q_re = torch.gather(query[:, None].expand(B, ncam, Q, E), 2, gi.expand(-1, -1, -1, E)) * valid[..., None]
        ref = reference_points_cam.permute(1, 0, 2, 3, 4)                      # (B,N,Q,Za,2)
        Za = ref.shape[3]
        r_re = torch.gather(ref, 2, gi[..., None].expand(-1, -1, -1, Za, 2)) * valid[..., None, None]
        d = bev_query_depth.permute(1, 0, 2, 3, 4).squeeze(-1)                 # (B,N,Q,Za)
        d_re = torch.gather(d, 2, gi.expand(-1, -1, -1, Za)) * valid[..., None]
        bins = torch.clip(torch.floor((d_re - self.dbound[0]) / self.dbound[2]), 0, DC - 1).to(torch.long)
        onehot = F.one_hot(bins, num_classes=DC)
        pred = pred_img_depth.reshape(B * ncam, DC, -1).permute(0, 2, 1)
        val = value.permute(2, 0, 1, 3).reshape(B * ncam, S, E)
        out = self.deformable_attention(query=q_re.reshape(B * ncam, max_len, E), key=val, value=val,
                                        reference_points=r_re.reshape(B * ncam, max_len, Za, 2),
                                        spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                        bev_query_depth=onehot.reshape(B * ncam, max_len, Za, DC),
                                        pred_img_depth=pred).view(B, ncam, max_len, E)
        out = out * valid[..., None]
        slots = torch.zeros_like(query)
        for i in range(ncam):                                                  # camera order == reference (:208-211)
            slots = slots.scatter_add(1, order[:, i, :, None].expand(-1, -1, E), out[:, i])
        count = torch.clamp(hit.sum(1), min=1.0)
        return slots / count[..., None]

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, reference_points_cam=None, level_start_index=None, flag='encoder',
                bev_query_depth=None, pred_img_depth=None, bev_mask=None, per_cam_mask_list=None, _defer_residual=False,
                bev_w=0, _norm=None, _ffn_tail=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        # inference on the fused route: `query + query_pos` is only consumed by the two projections of the sampler, whose
        # kernel adds the positional rows while it loads the query rows -- no pass of its own
        fold_pos = (query_pos is not None and self.fused and not torch.is_grad_enabled() and
                    x3_ok(query, query.shape[-1], 4))
        if query_pos is not None and not fold_pos:
            query = query + query_pos
        mask = per_cam_mask_list
        if bev_mask is not None:
            mask = mask & bev_mask[None, :, :, None]
        needs_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (query, value, pred_img_depth)) or \
            (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        head_dim = self.embed_dims // self.deformable_attention.num_heads
        fused_ok = self.fused and (not needs_grad or head_dim <= FUSED_BWD_MAX_HEAD_DIM)
        fn = self._slots_fused if fused_ok else self._slots_composite
        if fold_pos and fn != self._slots_fused:             # (cannot happen without autograd; kept for safety)
            query, fold_pos = query + query_pos, False
        kw = dict(query_pos=query_pos) if fold_pos else {}
        tail_ok = (_norm is not None and _defer_residual and self.layer_scale is None and not torch.is_grad_enabled()
                   and not (self.training and self.dropout.p > 0))
        tail = None
        if (tail_ok and FUSE_ATTN_TAIL_DA and fn == self._slots_fused and self.embed_dims % 16 == 0
                and _RL.ln_fusable(_norm, inp_residual.contiguous(), query, self.embed_dims)):
            # output_proj + residual + LayerNorm inside the sampler's (8-head) workgroups: fbbev_da_cross_attn_fused_ln (opt-in)
            if not hasattr(self.output_proj, '_x3'):
                self.output_proj._x3 = X3Weights()
            oc = self.output_proj._x3.get(self.output_proj.weight, self.output_proj.bias)
            tail = {'spec': (oc.frag, oc.b, inp_residual.contiguous(), _norm.weight, _norm.bias, _norm.eps), 'done': False}
            kw['_tail'] = tail
        slots = fn(query, value, reference_points_cam, mask, bev_query_depth, pred_img_depth, spatial_shapes,
                   level_start_index, bev_w=bev_w or 0, **kw)  # the BEV row length lets the sampler own 2-D patches of queries
        if tail is not None and tail['done']:
            return slots, _NORMED
        if tail_ok and _ffn_tail is not None and _RL.X3 and slots.is_cuda and slots.dtype == torch.float32:
            # round 5: output_proj + residual + norm AND the layer's FFN + residual + norm in ONE kernel (fbbev_rows_tail_ffn_x3):
            # the block's output rows stay in the wave's registers between the two
            E = self.embed_dims
            res = inp_residual.contiguous()
            c1, c2, H = _ffn_tail['spec']
            n1 = _ffn_tail['norm1']
            s2, r2 = slots.reshape(-1, E), res.reshape(-1, E)
            if (_RL.ln_fusable(_norm, res, slots, E) and _RL.ln_fusable(n1, None, slots, E) and self.output_proj.bias is not None
                    and _capi.rows_tail_ffn_x3_supported(s2, r2, E, H)):
                if not hasattr(self.output_proj, '_x3'):
                    self.output_proj._x3 = X3Weights()
                oc = self.output_proj._x3.get(self.output_proj.weight, self.output_proj.bias)
                planes = _ffn_tail.get('planes')
                if planes and s2.shape[0] % planes == 0 and _OUT_PLANES is not None:
                    out = _capi.rows_tail_ffn_x3(s2, oc.frag, oc.b, r2, _norm.weight, _norm.bias, _norm.eps, c1.frag, c1.b, c2.frag, c2.b, H,
                                                 n1.weight, n1.bias, n1.eps, tokens_per_image=planes)
                    _ffn_tail['done'] = True
                    _OUT_PLANES['got'] = True
                    return out, _NORMED                         # (B, C, Y * X): BackwardProjection.forward views it as (B, C, Y, X)
                out = _capi.rows_tail_ffn_x3(s2, oc.frag, oc.b, r2, _norm.weight, _norm.bias, _norm.eps, c1.frag, c1.b, c2.frag, c2.b, H,
                                             n1.weight, n1.bias, n1.eps)
                _ffn_tail['done'] = True
                return out.view(slots.shape), _NORMED
        if tail_ok:
            # output_proj + residual + the layer's following LayerNorm in one kernel (fbbev_rows_linear_x3_ln)
            return self.output_proj(slots, ln=(inp_residual.contiguous(), _norm)), _NORMED
        slots = self.output_proj(slots)
        if self.layer_scale is not None:
            slots = self.layer_scale * slots
        return (self.dropout(slots), inp_residual) if _defer_residual else self.dropout(slots) + inp_residual


class _LayerNormRows(torch.autograd.Function):
    """fbbev_layernorm / fbbev_layernorm_bwd under autograd (training of the backward projection on a GPU)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return _capi.layernorm(x, weight, bias, eps)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx, gw, gb = _capi.layernorm_bwd(x, gy.float().contiguous(), weight, ctx.eps)
        return (gx if ctx.needs_input_grad[0] else None, gw if ctx.needs_input_grad[1] else None,
                gb if ctx.needs_input_grad[2] else None, None)


class LayerNorm(nn.LayerNorm):
    """torch.nn.LayerNorm (= mmcv build_norm_layer('LN')) on fbbev_layernorm: half a wave64 per 80-float row instead of
    torch's generic kernel (150 us -> ~25 us on 160k rows), and under autograd fbbev_layernorm_bwd instead of ATen's three
    backward kernels (0.47 ms per LayerNorm at 160k rows).  Same parameters / state_dict; for shapes the kernels do not take
    (and on the CPU) it is exactly nn.LayerNorm."""

    def shape_ok(self, x):
        C = x.shape[-1]
        return (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and self.bias is not None and
                len(self.normalized_shape) == 1 and C % 4 == 0 and C <= 128 and x.is_contiguous())

    def wants_grad(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or self.bias.requires_grad)

    def kernel_ok(self, x):
        return self.shape_ok(x) and not self.wants_grad(x)

    def forward(self, x, residual=None):
        """LN(x [+ residual]); the inference kernel adds the residual while it reads the row (the same fp32 sum as a separate add)."""
        if self.kernel_ok(x) and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype
                                                       and residual.is_contiguous() and not
                                                       (torch.is_grad_enabled() and residual.requires_grad))):
            return _capi.layernorm(x, self.weight, self.bias, self.eps, residual=residual)
        if residual is not None:
            x = x + residual
        if self.shape_ok(x) and self.wants_grad(x) and x.shape[-1] == self.weight.numel():
            return _LayerNormRows.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)


@register
class BEVFormerEncoderLayer(nn.Module):
    """bevformer_encoder.py:206-377 (+ custom_base_transformer_layer.py:38-175)."""

    def __init__(self, attn_cfgs, feedforward_channels=512, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2, ffn_cfgs=None,
                 batch_first=True, **kwargs):
        super().__init__()
        assert len(operation_order) in {2, 4, 6}
        self.operation_order = tuple(operation_order)
        self.batch_first = batch_first
        self.pre_norm = operation_order[0] == 'norm'
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert len(attn_cfgs) == num_attn
        self.attentions = nn.ModuleList()
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            cfg.setdefault('batch_first', batch_first)
            self.attentions.append(build(cfg))
        self.embed_dims = self.attentions[0].embed_dims
        ffn = dict(type='FFN', embed_dims=self.embed_dims, feedforward_channels=feedforward_channels,
                   num_fcs=ffn_num_fcs, ffn_drop=ffn_dropout, act_cfg=act_cfg)
        if ffn_cfgs:
            ffn.update(ffn_cfgs)
            ffn['feedforward_channels'] = feedforward_channels     # deprecated-arg override (:88-99)
            ffn['ffn_drop'] = ffn_dropout
        self.ffns = nn.ModuleList([build(ffn) for _ in range(operation_order.count('ffn'))])
        self.norms = nn.ModuleList([LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, bev_pos=None, ref_2d=None, ref_3d=None, bev_h=None, bev_w=None,
                reference_points_cam=None, spatial_shapes=None, level_start_index=None, bev_mask=None,
                bev_query_depth=None, per_cam_mask_list=None, pred_img_depth=None, key_pos=None, **kwargs):
        if torch.is_grad_enabled() and query.is_cuda and value is not None and value.dim() == 4 and pred_img_depth is not None:
            # training (round 6): the whole layer as ONE autograd node on the inference kernels (train_path.EncoderLayerFn)
            from . import train_path as TP
            ncam, S_, bs_, E_ = value.shape
            rows = value.permute(2, 0, 1, 3).reshape(bs_ * ncam, S_, E_)
            if (TP.TRAIN_FUSED and rows.is_contiguous() and reference_points_cam is not None and bev_pos is not None and
                    TP.layer_supported(self, query, bev_pos, rows, pred_img_depth, reference_points_cam, spatial_shapes, bev_h, bev_w,
                                       bev_mask)):
                return TP.run_layer(self, query, bev_pos, rows, pred_img_depth, ref_2d, reference_points_cam, per_cam_mask_list,
                                    bev_query_depth, spatial_shapes, level_start_index, bev_h, bev_w)
        ni = ai = fi = 0
        identity = query
        ops = self.operation_order
        # post-norm inference: a branch followed by 'norm' hands (output, residual) to the LayerNorm kernel, which adds them
        # while it reads the row (three element-wise passes over the BEV queries less per layer; the same fp32 sums)
        defer = not self.pre_norm and not torch.is_grad_enabled() and query.is_cuda
        pending = None
        skip = 0
        for k, layer in enumerate(ops):
            if skip:                                      # ops a previous branch already ran inside its own kernel
                skip -= 1
                continue
            d = defer and k + 1 < len(ops) and ops[k + 1] == 'norm'
            nk = dict(_norm=self.norms[ni]) if d and FUSE_OUT_NORM else {}      # the branch may run its following norm itself
            if layer == 'self_attn':
                query = self.attentions[ai](
                    query, None, None, identity if self.pre_norm else None, query_pos=bev_pos, key_pos=bev_pos,
                    key_padding_mask=bev_mask, reference_points=ref_2d,
                    spatial_shapes=const_tensor([[bev_h, bev_w]], query.device),
                    level_start_index=const_tensor([0], query.device), _defer_residual=d, **nk)
                ai += 1
                if d:
                    query, pending = query
                identity = query
            elif layer == 'norm':
                if pending is not _NORMED:                    # _NORMED: norm ni already ran in the previous branch's last GEMM
                    query = self.norms[ni](query, pending)
                pending = None
                ni += 1
            elif layer == 'cross_attn':
                ft = None
                if nk and FUSE_TAIL_FFN and ops[k + 1:k + 4] == ('norm', 'ffn', 'norm') and fi < len(self.ffns):
                    # the block's tail and the following FFN block as one row kernel when the attention module can hand over its slots
                    spec = self.ffns[fi].fused_tail_spec(self.embed_dims) if hasattr(self.ffns[fi], 'fused_tail_spec') else None
                    if spec is not None:
                        ft = dict(spec=spec, norm1=self.norms[ni + 1], done=False)
                        if _OUT_PLANES is not None and _OUT_PLANES['last'] and k + 4 == len(ops):
                            ft['planes'] = _OUT_PLANES['tokens']      # the layer ends with this kernel: it may write (B, C, Y, X)
                        nk = dict(nk, _ffn_tail=ft)
                query = self.attentions[ai](
                    query, key, value, identity if self.pre_norm else None, query_pos=bev_pos, key_pos=key_pos,
                    reference_points=ref_3d, reference_points_cam=reference_points_cam, spatial_shapes=spatial_shapes,
                    level_start_index=level_start_index, bev_query_depth=bev_query_depth,
                    pred_img_depth=pred_img_depth, bev_mask=bev_mask, per_cam_mask_list=per_cam_mask_list,
                    _defer_residual=d, bev_w=bev_w, **nk)
                ai += 1
                if d:
                    query, pending = query
                identity = query
                if ft is not None and ft['done']:         # 'norm', 'ffn', 'norm' ran inside the attention block's tail kernel
                    skip, pending = 3, None
                    ni += 2
                    fi += 1
            elif layer == 'ffn':
                query = self.ffns[fi](query, identity if self.pre_norm else None, _defer_residual=d, **nk)
                fi += 1
                if d:
                    query, pending = query
        return query


@register
class bevformer_encoder(nn.Module):
    """bevformer_encoder.py:27-203."""

    def __init__(self, transformerlayers=None, num_layers=None, pc_range=None, grid_config=None, data_config=None,
                 return_intermediate=False, dataset_type='nuscenes', fix_bug=False, init_cfg=None, **kwargs):
        super().__init__()
        layers = transformerlayers if isinstance(transformerlayers, list) else \
            [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        self.layers = nn.ModuleList([build(c) for c in layers])
        self.num_layers = num_layers
        self.embed_dims = self.layers[0].embed_dims
        self.return_intermediate = return_intermediate
        self.x_bound, self.y_bound, self.z_bound = grid_config['x'], grid_config['y'], grid_config['z']
        self.final_dim = data_config['input_size']
        self.pc_range = pc_range

    def get_reference_points(self, H, W, Z=8, dim='3d', bs=1, device='cuda', dtype=torch.float):
        """:52-89."""
        if dim == '3d':
            X = torch.arange(*self.x_bound, dtype=torch.float) + self.x_bound[-1] / 2
            Y = torch.arange(*self.y_bound, dtype=torch.float) + self.y_bound[-1] / 2
            Zs = torch.arange(*self.z_bound, dtype=torch.float) + self.z_bound[-1] / 2
            Y, X, Zs = torch.meshgrid([Y, X, Zs], indexing='ij')
            return torch.stack([X, Y, Zs], dim=-1).to(dtype).to(device)
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device),
                                      torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device), indexing='ij')
        ref_y = ref_y.reshape(-1)[None] / H
        ref_x = ref_x.reshape(-1)[None] / W
        return torch.stack((ref_x, ref_y), -1).repeat(bs, 1, 1).unsqueeze(2)

    def _axes(self, device):
        """Voxel-centre axes of get_reference_points('3d'), cached per device."""
        if not hasattr(self, '_axes_cache'):
            self._axes_cache = {}
        if device not in self._axes_cache:
            ax = [torch.arange(*b, dtype=torch.float) + b[-1] / 2 for b in (self.x_bound, self.y_bound, self.z_bound)]
            self._axes_cache[device] = tuple(t.to(device).contiguous() for t in ax)
        return self._axes_cache[device]

    def point_sampling(self, reference_points, pc_range, img_metas, cam_params=None, gt_bboxes_3d=None):
        """:91-120 -- ego voxel centres -> per-camera normalised pixel coords, in-image mask, camera depth;
        one HIP kernel (fbbev_point_sampling).  `reference_points` must be the '3d' grid of this encoder."""
        rots, trans, intrins, post_rots, post_trans, bda = [t.float().contiguous() for t in cam_params]
        B, N, _ = trans.shape
        xs, ys, zs = self._axes(trans.device)
        Q, Za = ys.numel() * xs.numel(), zs.numel()
        ref_cam = torch.empty((N, B, Q, Za, 2), dtype=torch.float32, device=trans.device)
        mask = torch.empty((N, B, Q, Za), dtype=torch.bool, device=trans.device)
        qdepth = torch.empty((N, B, Q, Za), dtype=torch.float32, device=trans.device)
        ogfH, ogfW = self.final_dim
        _capi.point_sampling(xs, ys, zs, rots, trans, intrins, post_rots, post_trans, bda, ogfH, ogfW, ref_cam, mask,
                             qdepth)
        return reference_points, ref_cam, mask, qdepth.unsqueeze(-1)

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None, spatial_shapes=None,
                level_start_index=None, cam_params=None, gt_bboxes_3d=None, pred_img_depth=None, bev_mask=None,
                prev_bev=None, **kwargs):
        """:123-203."""
        ck = (bev_h, bev_w, bev_query.size(1), bev_query.device, bev_query.dtype)
        if getattr(self, '_ref_cache_key', None) != ck:        # config-only tensors: build once
            self._ref_cache = (
                self.get_reference_points(bev_h, bev_w, self.pc_range[5] - self.pc_range[2], dim='3d',
                                          bs=bev_query.size(1), device=bev_query.device, dtype=bev_query.dtype),
                self.get_reference_points(bev_h, bev_w, dim='2d', bs=bev_query.size(1), device=bev_query.device,
                                          dtype=bev_query.dtype))
            self._ref_cache_key = ck
        ref_3d, ref_2d = self._ref_cache
        if _PRE is not None and _PRE.sampling is not None and _PRE.sampling_key == _tkey(cam_params):
            ref_cam, per_cam_mask_list, bev_query_depth = _PRE.sampling          # sampled on the side stream (prefetch)
        else:
            ref_3d, ref_cam, per_cam_mask_list, bev_query_depth = self.point_sampling(
                ref_3d, self.pc_range, kwargs.get('img_metas'), cam_params=cam_params, gt_bboxes_3d=gt_bboxes_3d)
        bev_query = bev_query.permute(1, 0, 2)
        bev_pos = bev_pos.permute(1, 0, 2)
        output, inter = bev_query, []
        for li, layer in enumerate(self.layers):
            if _OUT_PLANES is not None:
                _OUT_PLANES['last'] = li == len(self.layers) - 1 and not self.return_intermediate
            output = layer(bev_query, key, value, bev_pos=bev_pos, ref_2d=ref_2d, ref_3d=ref_3d, bev_h=bev_h,
                           bev_w=bev_w, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                           reference_points_cam=ref_cam, per_cam_mask_list=per_cam_mask_list, bev_mask=bev_mask,
                           bev_query_depth=bev_query_depth, pred_img_depth=pred_img_depth)
            bev_query = output
            if self.return_intermediate:
                inter.append(output)
        return torch.stack(inter) if self.return_intermediate else output


@register
class BEVFormer(nn.Module):
    """bevformer.py:22-132."""

    def __init__(self, num_cams=6, encoder=None, embed_dims=256, output_dims=256, use_cams_embeds=True, **kwargs):
        super().__init__()
        self.encoder = build(encoder)
        self.embed_dims, self.num_cams, self.output_dims = embed_dims, num_cams, output_dims
        self.use_cams_embeds = use_cams_embeds
        self.fused_tokens = True
        self.cams_embeds = nn.Parameter(torch.Tensor(num_cams, embed_dims))
        self.init_weights()

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (DA_MSDeformableAttention, MultiScaleDeformableAttention)):
                m.init_weights()
        nn.init.normal_(self.cams_embeds)

    def _tokens_fusable(self, mlvl_feats):
        f0 = mlvl_feats[0]
        needs_grad = torch.is_grad_enabled() and (self.cams_embeds.requires_grad or any(f.requires_grad for f in mlvl_feats))
        return (self.fused_tokens and f0.is_cuda and f0.dtype == torch.float32 and not needs_grad and f0.shape[1] == self.num_cams
                and all(f.shape[:3] == f0.shape[:3] for f in mlvl_feats))

    def _tokens_trainable(self, mlvl_feats):
        from . import train_path as TP
        f0 = mlvl_feats[0]
        return (TP.TRAIN_FUSED and torch.is_grad_enabled() and self.fused_tokens and f0.is_cuda and f0.dtype == torch.float32 and
                f0.shape[1] == self.num_cams and all(f.shape[:3] == f0.shape[:3] and f.dtype == torch.float32 for f in mlvl_feats))

    def _token_rows(self, mlvl_feats, shapes):
        """(bs * num_cam, sum HW, C) camera-token rows: feat.flatten(3).permute + cams_embeds + cat + the rebatch permute of the
        reference (bevformer.py:95-117, spatial_cross_attention_depth.py:151) in one transposing launch"""
        f0 = mlvl_feats[0]
        bs, ncam, c = f0.shape[:3]
        S = sum(h * w for h, w in shapes)
        rows = torch.empty((bs * ncam, S, c), dtype=torch.float32, device=f0.device)
        ck = (self.cams_embeds.data_ptr(), self.cams_embeds._version, str(f0.device), self.use_cams_embeds)
        if getattr(self, '_ce_key', None) != ck:       # per-weight-version constant (the reference adds `cams_embeds * 0` when unused)
            ce = self.cams_embeds.detach().to(torch.float32)
            self._ce, self._ce_key = (ce if self.use_cams_embeds else ce * 0).contiguous(), ck
        ce = self._ce
        if 1 < len(mlvl_feats) <= 8:       # the whole pyramid in one launch
            _capi.tokens_from_nchw_levels([f.reshape(bs * ncam, c, h * w).contiguous() for f, (h, w) in zip(mlvl_feats, shapes)], rows, ce)
        else:
            start = 0
            for feat, (h, w) in zip(mlvl_feats, shapes):
                _capi.tokens_from_nchw(feat.reshape(bs * ncam, c, h * w).contiguous(), rows, start * c, ce)
                start += h * w
        return rows

    def forward(self, mlvl_feats, bev_queries, bev_h, bev_w, bev_pos=None, cam_params=None, gt_bboxes_3d=None,
                pred_img_depth=None, prev_bev=None, bev_mask=None, **kwargs):
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        shapes = [tuple(feat.shape[-2:]) for feat in mlvl_feats]
        f0 = mlvl_feats[0]
        if self._tokens_fusable(mlvl_feats):
            # inference: one transposing pass per level (fbbev_tokens_from_nchw) writes feat + cams_embeds straight into the
            # (bs*num_cam, sum HW, C) token rows the cross-attention's value projection reads -- the flatten/permute/add,
            # the cat and the rebatch permute of the reference (three copies of the camera features) in one.  What the
            # encoder receives is the reference's (num_cam, sum HW, bs, C) tensor as a VIEW of those rows.
            bs, ncam, c = f0.shape[:3]
            S = sum(h * w for h, w in shapes)
            rows = _PRE.rows if (_PRE is not None and _PRE.rows is not None and _PRE.rows_key == _tkey(mlvl_feats)) else None
            if rows is None:
                rows = self._token_rows(mlvl_feats, shapes)
            feat_flatten = rows.view(bs, ncam, S, c).permute(1, 0, 2, 3)            # (num_cam, bs, sum HW, C)
        elif self._tokens_trainable(mlvl_feats):
            # training (round 6): the same one-launch token rows as a differentiable op (train_path.TokenRows)
            from . import train_path as TP
            bs, ncam, c = f0.shape[:3]
            S = sum(h * w for h, w in shapes)
            rows = TP.TokenRows.apply(self.cams_embeds, self.use_cams_embeds, *mlvl_feats)
            feat_flatten = rows.view(bs, ncam, S, c).permute(1, 0, 2, 3)
        else:
            feats = []
            for feat in mlvl_feats:
                f = feat.flatten(3).permute(1, 0, 3, 2)
                ce = self.cams_embeds[:, None, None, :].to(f.dtype)
                feats.append(f + (ce if self.use_cams_embeds else ce * 0))
            feat_flatten = torch.cat(feats, 2)
        if pred_img_depth is not None and tuple(pred_img_depth.shape[-2:]) != tuple(shapes[0]):
            # the depth distribution is sampled on spatial_shapes[0:1] (spatial_cross_attention_depth.py:586):
            # level 0 must be the level the depth net ran on, or the sampling would index past the depth map
            raise ValueError(f'pred_img_depth is {tuple(pred_img_depth.shape[-2:])} but feature level 0 is {tuple(shapes[0])}')
        spatial_shapes = const_tensor(shapes, bev_pos.device)
        starts = [0]
        for h, w in shapes[:-1]:
            starts.append(starts[-1] + h * w)
        level_start_index = const_tensor(starts, bev_pos.device)
        feat_flatten = feat_flatten.permute(0, 2, 1, 3)                        # (num_cam, sum HW, bs, C)
        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                            spatial_shapes=spatial_shapes, level_start_index=level_start_index, cam_params=cam_params,
                            gt_bboxes_3d=gt_bboxes_3d, pred_img_depth=pred_img_depth, prev_bev=prev_bev,
                            bev_mask=bev_mask, **kwargs)


@register
class BackwardProjection(nn.Module):
    """backward_projection.py:34-133."""

    def __init__(self, *args, transformer=None, positional_encoding=None, pc_range=None, in_channels=64,
                 out_channels=64, use_zero_embedding=False, bev_h=30, bev_w=30, **kwargs):
        super().__init__()
        self.bev_h, self.bev_w, self.pc_range = bev_h, bev_w, pc_range
        self.use_zero_embedding = use_zero_embedding
        self.real_w = pc_range[3] - pc_range[0]
        self.real_h = pc_range[4] - pc_range[1]
        self.positional_encoding = build(positional_encoding)
        self.transformer = build(transformer)
        self.embed_dims = self.transformer.embed_dims
        self.bev_embedding = nn.Embedding(bev_h * bev_w, self.embed_dims)

    def init_weights(self):
        self.transformer.init_weights()

    def prefetch(self, mlvl_feats, cam_params):
        """Inference: start everything of the backward projection that does not need `lss_bev` -- the camera-token rows, their value
        projection as head planes (first encoder layer), the BEV -> image point sampling -- on a side stream, so that it runs under the
        caller's latency-bound ranking / Z-mean kernels.  Returns a handle for `forward(..., _pre=handle)` (same input tensors), or
        None when the route does not apply (autograd, hipGraph capture, CPU, FBBEV_BP_PREFETCH=0)."""
        f0 = mlvl_feats[0]
        tr = self.transformer
        if (not PREFETCH or torch.is_grad_enabled() or not f0.is_cuda or cam_params is None or not isinstance(tr, BEVFormer)
                or torch.cuda.is_current_stream_capturing() or not tr._tokens_fusable(mlvl_feats)):
            return None
        # measured (profiles/r05_time_fb_prefetch.jsonl): -18 us at BASELINE configs[2] B = 4 (160 000 queries: the step is GPU-bound),
        # but +80-95 us at the shipped shape B = 1 / 4 and at configs[2] B = 1, where the eager step is bound by the host's launches and
        # the stream switch + event cost host time: only for large launches (a captured hipGraph never takes this route)
        if f0.shape[0] * self.bev_h * self.bev_w < PREFETCH_MIN_QUERIES:
            return None
        main = torch.cuda.current_stream(f0.device)
        if getattr(self, '_side_dev', None) != f0.device:
            self._side, self._side_dev = torch.cuda.Stream(f0.device), f0.device
        side = self._side
        side.wait_stream(main)                                   # the inputs were produced on the caller's stream
        pre = _Prefetch()
        keep = []
        with torch.cuda.stream(side):
            shapes = [tuple(f.shape[-2:]) for f in mlvl_feats]
            rows = tr._token_rows(mlvl_feats, shapes)
            pre.rows, pre.rows_key = rows, _tkey(mlvl_feats)
            keep.append(rows)
            enc = tr.encoder
            BN, S, E = rows.shape
            bs = f0.shape[0]
            if _RL.X3 and E % 8 == 0 and min(w for _, w in shapes) >= 2:
                za = enc._axes(f0.device)[2].numel()
                for att in enc.layers[0].attentions:
                    if isinstance(att, DA_SpatialCrossAttention) and att.value_dtype is None and att.fused:
                        da = att.deformable_attention
                        if (not da.disable_deformable and E == att.embed_dims and _capi.da_cross_attn_fused_supported(
                                bs, BN // bs, S, da.num_heads, E // da.num_heads, da.num_levels, self.bev_h * self.bev_w,
                                da.num_points, za, self.bev_w)):
                            planes = att._value_planes(rows.reshape(BN * S, E), S)
                            pre.planes[id(att)] = ((rows.data_ptr(), BN, S, E), planes)
                            keep.append(planes)
            _, ref_cam, mask, qd = enc.point_sampling(None, enc.pc_range, None, cam_params=cam_params)
            pre.sampling, pre.sampling_key = (ref_cam, mask, qd), _tkey(cam_params)
            keep += [ref_cam, mask, qd]
            pre.event = torch.cuda.Event()
            pre.event.record(side)
        for t in keep:
            t.record_stream(main)                                # allocated on the side stream, consumed on the caller's
        return pre

    def query_row_bias(self, channels, grid_zyx):
        """bev_embedding as the (Y*X, C) row bias of fbbev_pool_zmean_rows, or None when the queries cannot be handed over as rows
        (grad mode, another grid / width than the module's)."""
        w = self.bev_embedding.weight
        if (torch.is_grad_enabled() and w.requires_grad) or not w.is_cuda or w.dtype != torch.float32:
            return None
        if tuple(w.shape) != (self.bev_h * self.bev_w, channels) or (grid_zyx[1], grid_zyx[2]) != (self.bev_h, self.bev_w):
            return None
        return w.detach().contiguous()

    def forward(self, mlvl_feats, img_metas, lss_bev=None, gt_bboxes_3d=None, cam_params=None, pred_img_depth=None,
                bev_mask=None, _pre=None, lss_rows=None):
        """lss_rows (inference): the queries as rows (bs, Y*X, C) ALREADY carrying bev_embedding (fbbev_pool_zmean_rows) instead of
        lss_bev (bs, C, Y, X)."""
        global _PRE
        if _pre is not None:
            torch.cuda.current_stream(mlvl_feats[0].device).wait_event(_pre.event)
            _PRE = _pre
            try:
                return self.forward(mlvl_feats, img_metas, lss_bev=lss_bev, gt_bboxes_3d=gt_bboxes_3d, cam_params=cam_params,
                                    pred_img_depth=pred_img_depth, bev_mask=bev_mask, lss_rows=lss_rows)
            finally:
                _PRE = None
        bs = mlvl_feats[0].shape[0]
        dtype = mlvl_feats[0].dtype
        if lss_rows is not None:
            assert lss_bev is None and lss_rows.is_cuda and lss_rows.dtype == torch.float32 and not torch.is_grad_enabled()
            lss_bev = lss_rows                                  # (takes the `fast` route below; used only for its device / dtype / grad state)
        # (Q,bs,C) as backward_projection.py:96-99 -- built batch-major so that the encoder's permute(1,0,2) yields
        # contiguous (bs,Q,C) tokens (the Linear layers then take them without a copy); same sums element for element
        fast = (lss_bev is not None and lss_bev.is_cuda and lss_bev.dtype == torch.float32 and
                not (torch.is_grad_enabled() and (lss_bev.requires_grad or self.bev_embedding.weight.requires_grad)))
        from . import train_path as TP
        train_fast = (not fast and TP.TRAIN_FUSED and torch.is_grad_enabled() and lss_bev is not None and lss_bev.is_cuda and
                      lss_bev.dtype == torch.float32 and dtype == torch.float32)
        if lss_bev is not None:
            if train_fast:                                     # the same transposing pass, differentiable (train_path.BevQueries)
                bev_queries = TP.BevQueries.apply(lss_bev, self.bev_embedding.weight).permute(1, 0, 2)
            elif lss_rows is not None:                                                  # rows + bev_embedding straight from the Z-mean
                bev_queries = lss_rows.permute(1, 0, 2)
            elif fast:                                                                  # LDS-tiled transposition kernel
                # + bev_embedding in the same pass: the same single fp32 add per element as below
                tok = _capi.tokens_from_nchw(lss_bev.reshape(bs, lss_bev.shape[1], -1).contiguous(),
                                             torch.empty((bs, self.bev_h * self.bev_w, lss_bev.shape[1]),
                                                         dtype=torch.float32, device=lss_bev.device),
                                             0, None, pos_bias=self.bev_embedding.weight.detach().float().contiguous())
                bev_queries = tok.permute(1, 0, 2)
            else:
                tok = lss_bev.flatten(2).transpose(1, 2).contiguous()                   # (bs,Q,C): the one transposition
                bev_queries = (tok + self.bev_embedding.weight.to(dtype).unsqueeze(0)).permute(1, 0, 2)
        else:
            bev_queries = self.bev_embedding.weight.to(dtype).unsqueeze(0).repeat(bs, 1, 1).permute(1, 0, 2)
        if bev_mask is not None:
            bev_mask = bev_mask.reshape(bs, -1)
        bev_pos = self.positional_encoding(bs, self.bev_h, self.bev_w, bev_queries.device).to(dtype)
        global _OUT_PLANES
        ctx = dict(tokens=self.bev_h * self.bev_w, last=False, got=False) if (fast and OUT_PLANES and not torch.is_grad_enabled()) else None
        _OUT_PLANES = ctx
        try:
            bev = self.transformer(mlvl_feats, bev_queries, self.bev_h, self.bev_w,
                                   grid_length=(self.real_h / self.bev_h, self.real_w / self.bev_w), bev_pos=bev_pos,
                                   img_metas=img_metas, cam_params=cam_params, gt_bboxes_3d=gt_bboxes_3d,
                                   pred_img_depth=pred_img_depth, prev_bev=None, bev_mask=bev_mask)
        finally:
            _OUT_PLANES = None
        if ctx is not None and ctx['got']:                     # the last layer's kernel wrote (bs, C, Y * X) itself
            return bev.view(bs, -1, self.bev_h, self.bev_w)
        if fast and bev.is_contiguous() and not bev.requires_grad:
            return _capi.transpose_last2(bev).view(bs, -1, self.bev_h, self.bev_w)
        if train_fast and bev.is_cuda and bev.dtype == torch.float32 and bev.dim() == 3:
            return TP.RowsToNCHW.apply(bev, self.bev_h, self.bev_w)
        return bev.permute(0, 2, 1).view(bs, -1, self.bev_h, self.bev_w).contiguous()
