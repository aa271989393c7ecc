"""Training step of the forward-backward view transformation on the kernels of its inference step (round 6).

The reference trains this path (bev_pool.py:40-80 / bev_pool_cuda.cu:64-118 for the lift-splat, multi_scale_deformable_attn_function.py:
137-172 for both attentions, autograd through every nn.Linear / LayerNorm of bevformer_encoder.py:206-377).  Until round 5 the training
forward here ran the round-3 kernels + vendor fp32 GEMMs + ~40 ATen launches (3.1 ms at BASELINE configs[2], B = 4, against 1.2 ms for
the same arithmetic in inference) and the backward another ~7 ms.  This module makes the encoder layer ONE autograd node:

  forward  = the inference route (query rows -> attention output in one kernel per block, tail + FFN in one row kernel), saving only
             the block boundaries (query rows, y0, slots, the two head-plane tensors) -- no offsets / attention-weight tensors;
  backward = recompute what a block needs from its saved inputs (projections on fbbev_rows_linear_x3, softmax), then the existing
             gradient kernels (fbbev_da_cross_attn_bwd_ws_grid, fbbev_msda_bwd_ws, fbbev_layernorm_bwd), every dgrad on the split-operand
             MFMA kernel with the transposed weight's fragments, every wgrad as a split-K batched GEMM.

and the volume is written once in training too (`WriteOnce`): the Z-mean comes from the index tensors, the refined BEV is added in the
pooling kernel's store epilogue, and the pooling backward runs ONCE, after the Z-mean's gradient has arrived (folded into its read of
the volume gradient).  FBBEV_TRAIN_FUSED=0 restores the round-5 composite route (A/B knob; the gradient tests run both).
"""
import os

import torch

from . import _capi
from . import rows_linear as _RL
from .rows_linear import X3Weights

TRAIN_FUSED = os.environ.get('FBBEV_TRAIN_FUSED', '1') != '0'
WGRAD_X3 = os.environ.get('FBBEV_TRAIN_WGRAD', '1') != '0'
DA_BWD_PLANES = os.environ.get('FBBEV_TRAIN_DA_PLANES', '1') != '0'    # A/B knob: the DA backward on the forward's head planes, outputs written in full
ORDER = ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')


# ------------------------------------------------------------------------------------------------ small differentiable re-layouts
class BevQueries(torch.autograd.Function):
    """(B, C, H, W) Z-mean + bev_embedding (Q, C) -> (B, Q, C) query rows (backward_projection.py:96-99) in one transposing pass."""

    @staticmethod
    def forward(ctx, lss_bev, emb):
        B, C = lss_bev.shape[:2]
        ctx.hw = tuple(lss_bev.shape[2:])
        tok = torch.empty((B, lss_bev.shape[2] * lss_bev.shape[3], C), dtype=torch.float32, device=lss_bev.device)
        return _capi.tokens_from_nchw(lss_bev.reshape(B, C, -1).contiguous(), tok, 0, None, pos_bias=emb.detach().float().contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        g_lss = _capi.transpose_last2(g).view(g.shape[0], g.shape[2], *ctx.hw) if ctx.needs_input_grad[0] else None
        return g_lss, (g.sum(0) if ctx.needs_input_grad[1] else None)


class RowsToNCHW(torch.autograd.Function):
    """(B, Q, C) rows -> (B, C, H, W) (backward_projection.py:127-130)"""

    @staticmethod
    def forward(ctx, rows, h, w):
        return _capi.transpose_last2(rows.contiguous()).view(rows.shape[0], rows.shape[2], h, w)

    @staticmethod
    def backward(ctx, g):
        B, C, h, w = g.shape
        return _capi.transpose_last2(g.reshape(B, C, h * w).contiguous()), None, None


class TokenRows(torch.autograd.Function):
    """list of (B, N, C, h_l, w_l) camera features -> (B * N, sum h_l w_l, C) token rows (+ cams_embeds): bevformer.py:95-117 and the
    rebatch permute of spatial_cross_attention_depth.py:151 in one launch (fbbev_tokens_from_nchw_levels)."""

    @staticmethod
    def forward(ctx, ce, use_ce, *feats):
        f0 = feats[0]
        B, N, C = f0.shape[:3]
        shapes = [tuple(f.shape[-2:]) for f in feats]
        S = sum(h * w for h, w in shapes)
        rows = torch.empty((B * N, S, C), dtype=torch.float32, device=f0.device)
        bias = (ce.detach().float() if use_ce else ce.detach().float() * 0).contiguous()
        lv = [f.reshape(B * N, C, h * w).contiguous() for f, (h, w) in zip(feats, shapes)]
        if 1 < len(lv) <= 8:
            _capi.tokens_from_nchw_levels(lv, rows, bias)
        else:
            start = 0
            for t, (h, w) in zip(lv, shapes):
                _capi.tokens_from_nchw(t, rows, start * C, bias)
                start += h * w
        ctx.shapes, ctx.bn, ctx.use_ce = shapes, (B, N, C), use_ce
        return rows

    @staticmethod
    def backward(ctx, g):
        B, N, C = ctx.bn
        g = g.contiguous()
        out, start = [], 0
        for k, (h, w) in enumerate(ctx.shapes):
            if ctx.needs_input_grad[2 + k]:
                out.append(_capi.transpose_last2(g[:, start:start + h * w].contiguous()).view(B, N, C, h, w))
            else:
                out.append(None)
            start += h * w
        g_ce = None
        if ctx.needs_input_grad[0]:      # the reference adds `cams_embeds * 0` when unused (bevformer.py:103): a zero gradient, not None
            g_ce = g.view(B, N, -1, C).sum((0, 2)) if ctx.use_ce else torch.zeros((N, C), dtype=g.dtype, device=g.device)
        return (g_ce, None, *out)


# ------------------------------------------------------------------------------------------------ volume written once, training
class WriteOnce:
    """Z-mean from the index tensors + `refined[..., None] + volume` in the pooling store (fbocc.py:344-366) as two autograd nodes that
    share ONE pooling backward: `PoolAdd.backward` hands the Z-sum of the output gradient to `refined` and parks the gradient;
    `ZMean.backward` -- which autograd can only reach through `refined`'s graph, i.e. later -- runs fbbev_bev_pool_v2_dense_bwd_z on the
    parked gradient with the mean's gradient folded in."""

    class ZMean(torch.autograd.Function):
        @staticmethod
        def forward(ctx, context, depth, fp, parts, shared):
            ctx.fp, ctx.parts, ctx.shared = fp, parts, shared
            ctx.set_materialize_grads(False)
            return fp.pooled_zmean(parts)

        @staticmethod
        def backward(ctx, g_mean):
            sh = ctx.shared
            og = sh.pop('g_out', None)
            if og is None and g_mean is None:
                return (None,) * 5
            return (*_pool_backward(ctx.fp, ctx.parts, og, g_mean), None, None, None)

    class PoolAdd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, context, depth, refined, fp, parts, shared):
            ctx.fp, ctx.parts, ctx.shared = fp, parts, shared
            return fp.pooled_volume(parts, addend=refined)

        @staticmethod
        def backward(ctx, g):
            g_ref = None
            if ctx.needs_input_grad[2]:
                if _capi.volume_zreduce_supported(g):          # the gradient has the volume's own memory layout
                    g_ref = _capi.volume_zreduce(g, 1.0)
                elif _capi.volume_zlast_supported(g):          # ... or is contiguous in the output's (B,C,Y,X,Z) shape
                    g_ref = _capi.volume_zreduce_inner(g, 1.0)
                else:
                    g_ref = g.sum(-1)
            og = g.permute(0, 1, 4, 2, 3)
            if ctx.needs_input_grad[2] and ctx.shared.get('defer', False):
                ctx.shared['g_out'] = og                       # context / depth gradients: by ZMean.backward, with the mean's folded in
                return None, None, g_ref, None, None, None
            gc, gd = _pool_backward(ctx.fp, ctx.parts, og, None)
            return gc, gd, g_ref, None, None, None


def _pool_backward(fp, parts, og, g_mean):
    """-> (grad context (B,N,C,H,W) view, grad depth)"""
    idx, depth, feat, _ = parts
    Z, Y, X = fp.grid_zyx
    B, N, D, H, W = depth.shape
    C = feat.shape[-1]
    if og is None:
        og = torch.zeros((B, C, Z, Y, X), dtype=torch.float32, device=depth.device)
    sb, sc = og.stride(0), og.stride(1)
    if (og.dtype != torch.float32 or og.stride()[2:] != (Y * X, X, 1) or sc < Z * Y * X or sb < C * sc or sc % 4 or sb % 4
            or og.data_ptr() % 16):
        zl = og.permute(0, 1, 3, 4, 2)
        og = _capi.volume_z_to_front(zl) if _capi.volume_zlast_supported(zl) else og.contiguous().float()
    ws = fp._ws.bwd_workspace(depth.device, _capi.pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X))
    dg, fg = torch.empty_like(depth), torch.empty_like(feat)
    zg = None if g_mean is None else g_mean.contiguous().float()
    _capi.bev_pool_v2_dense_bwd(og, depth, feat, idx.ranks_depth, idx.interval_rank, idx.interval_starts, idx.counts, idx.n,
                                (Z, Y, X), dg, fg, ws, zgrad=zg, zscale=1.0 / Z)
    return fg.permute(0, 1, 4, 2, 3), dg


def write_once_supported(fp, context):
    Z, Y, X = fp.grid_zyx
    return (fp.fused and not fp.extra_relu and fp.out_dtype == torch.float32 and context.is_cuda and context.dtype == torch.float32
            and fp._fused_supported(context.shape[2]) and (Y * X) % 4 == 0)


# ------------------------------------------------------------------------------------------------ the encoder layer as one node
def _frag(layer, name, w, b=None, transform=None):
    """split-operand fragments of a (derived) weight matrix, cached on the layer per source version"""
    cache = layer.__dict__.setdefault('_train_x3', {})
    if name not in cache:
        cache[name] = X3Weights()
    return cache[name].get(w, b, transform)


def _t(w_, b_):
    return w_.t().contiguous(), None


def _lin(x2d, c, relu=False, addend=None, out=None, res=None, mask=None):
    if res is None and mask is None:
        return _capi.rows_linear_x3(x2d, c.frag, c.b, c.w.shape[0], relu=relu, addend=addend, out=out)
    return _capi.rows_linear_x3_train(x2d, c.frag, c.b, c.w.shape[0], relu=relu, addend=addend, residual=res, mask=mask, out=out)


def layer_supported(layer, query, bev_pos, value_rows, pred_depth, reference_points_cam, spatial_shapes, bev_h, bev_w, bev_mask):
    from . import backward_projection as BP
    if not (TRAIN_FUSED and _RL.X3 and query.is_cuda and query.dtype == torch.float32 and bev_mask is None):
        return False
    if tuple(layer.operation_order) != ORDER or layer.pre_norm or len(layer.attentions) != 2 or len(layer.ffns) != 1:
        return False
    sa, ca = layer.attentions
    if not (isinstance(sa, BP.MultiScaleDeformableAttention) and isinstance(ca, BP.DA_SpatialCrossAttention)):
        return False
    da = ca.deformable_attention
    ffn = layer.ffns[0]
    E = layer.embed_dims
    if layer.training and (sa.dropout.p > 0 or ca.dropout.p > 0 or ffn._has_live_dropout()):
        return False
    if not (sa.batch_first and sa.num_levels == 1 and ca.fused and ca.layer_scale is None and ca.value_dtype is None and
            not da.disable_deformable and da.batch_first and sa.embed_dims == E and ca.embed_dims == E and E % 16 == 0):
        return False
    spec = ffn.fused_tail_spec(E)
    if spec is None:
        return False
    B, Q, _ = query.shape
    if Q != bev_h * bev_w or bev_pos is None:
        return False
    M, Dh = sa.num_heads, E // sa.num_heads
    hw = BP.host_values(spatial_shapes)
    if hw is None or min(int(w) for _, w in hw) < 2 or da.num_heads != M:
        return False
    BN, S, _ = value_rows.shape
    Za = reference_points_cam.shape[3]
    if not (_capi.msda_self_fused_supported(B, Q, M, Dh, 1, Q, sa.num_points, bev_w) and
            _capi.da_cross_attn_fused_supported(B, BN // B, S, M, Dh, da.num_levels, Q, da.num_points, Za, bev_w)):
        return False
    HS = (Dh + 3) // 4 * 4
    if _capi.da_cross_attn_bwd_ws_bytes(B, BN // B, S, M, Dh, Q, HS, da.num_levels, da.num_points, hw) <= 0:
        return False
    if _capi.msda_bwd_ws_bytes(B, Q, M, Dh, 1, Q, sa.num_points, [(bev_h, bev_w)]) <= 0:
        return False
    n0, n1, n2 = layer.norms
    for n in (n0, n1, n2):
        if not (_RL.ln_fusable(n, None, query, E) and n.weight.data_ptr() % 16 == 0 and n.bias.data_ptr() % 16 == 0):
            return False
    return True


def layer_params(layer):
    sa, ca = layer.attentions
    da = ca.deformable_attention
    l1, l2 = [l for l in layer.ffns[0].layers if not isinstance(l, torch.nn.Dropout)]
    l1 = l1[0]
    n0, n1, n2 = layer.norms
    mods = (sa.value_proj, sa.sampling_offsets, sa.attention_weights, sa.output_proj, n0,
            da.value_proj, da.sampling_offsets, da.attention_weights, ca.output_proj, n1, l1, l2, n2)
    out = []
    for m in mods:
        out += [m.weight, m.bias]
    return out


NAMES = ('sv', 'sso', 'saw', 'so', 'n0', 'cv', 'cso', 'caw', 'co', 'n1', 'f1', 'f2', 'n2')


class EncoderLayerFn(torch.autograd.Function):
    """BEVFormerEncoderLayer (bevformer_encoder.py:206-377: self_attn, norm, cross_attn, norm, ffn, norm) as one autograd node."""

    @staticmethod
    def forward(ctx, layer, geo, q, pos, rows, depth, *params):
        p = dict(zip(NAMES, zip(params[0::2], params[1::2])))
        sa, ca = layer.attentions
        da = ca.deformable_attention
        B, Q, E = q.shape
        M, Dh = sa.num_heads, E // sa.num_heads
        q = q.contiguous()
        q2 = q.view(B * Q, E)
        pos = pos.contiguous()
        # ---- self-attention block: value planes, then query rows -> LayerNorm(output_proj(attention) + q) in one kernel
        c = _frag(layer, 'sv', *p['sv'])
        planes_s = _capi.rows_linear_x3_planes(q2, c.frag, c.b, Q, M, Dh)
        cso, caw, co = _frag(layer, 'sso', *p['sso']), _frag(layer, 'saw', *p['saw']), _frag(layer, 'so', *p['so'])
        n0w, n0b = p['n0']
        y0 = torch.empty_like(q)
        _capi.msda_self_fused(planes_s, geo['ref2d'], q, pos, cso.frag, cso.b, caw.frag, caw.b, sa.num_points, geo['bev_w'],
                              (geo['bev_h'], geo['bev_w']), y0, out_proj=(co.frag, co.b, q, n0w, n0b, layer.norms[0].eps))
        # ---- cross-attention: camera-token head planes, then query rows -> slots in one kernel
        BN, S, _ = rows.shape
        c = _frag(layer, 'cv', *p['cv'])
        planes_c = _capi.rows_linear_x3_planes(rows.reshape(BN * S, E), c.frag, c.b, S, M, Dh)
        cso, caw = _frag(layer, 'cso', *p['cso']), _frag(layer, 'caw', *p['caw'])
        slots = torch.empty_like(q)
        _capi.da_cross_attn_fused(planes_c, geo['ss'], geo['ls'], depth, geo['ref_cam'], geo['mask'], geo['qdepth'], y0, pos,
                                  cso.frag, cso.b, caw.frag, caw.b, da.num_points, ca.dbound[0], ca.dbound[2], geo['bev_w'],
                                  geo['min_w'], slots)
        # ---- output_proj + residual + norm, FFN + residual + norm: one row kernel
        co, c1, c2 = _frag(layer, 'co', *p['co']), _frag(layer, 'f1', *p['f1']), _frag(layer, 'f2', *p['f2'])
        n1w, n1b = p['n1']
        n2w, n2b = p['n2']
        H = c1.w.shape[0]
        y2 = _capi.rows_tail_ffn_x3(slots.view(B * Q, E), co.frag, co.b, y0.view(B * Q, E), n1w, n1b, layer.norms[1].eps,
                                    c1.frag, c1.b, c2.frag, c2.b, H, n2w, n2b, layer.norms[2].eps).view(B, Q, E)
        ctx.layer, ctx.geo = layer, geo
        ctx.save_for_backward(q, pos, rows, depth, y0, slots, planes_s, planes_c, *params)
        return y2

    @staticmethod
    def backward(ctx, g_y2):
        layer, geo = ctx.layer, ctx.geo
        q, pos, rows, depth, y0, slots, planes_s, planes_c, *params = ctx.saved_tensors
        p = dict(zip(NAMES, zip(params[0::2], params[1::2])))
        need = dict(zip(NAMES, zip(ctx.needs_input_grad[6::2], ctx.needs_input_grad[7::2])))
        sa, ca = layer.attentions
        da = ca.deformable_attention
        B, Q, E = q.shape
        R = B * Q
        M, Dh = sa.num_heads, E // sa.num_heads
        dev = q.device
        G = {}                                        # name -> [grad weight, grad bias]

        def wgrad(name, gy, x, post=None, addend=None):
            # grad_weight = gy^T x and grad_bias = column sums of gy in one split-K MFMA kernel + a fixed-order reduction
            # (fbbev_rows_wgrad_x3); FBBEV_TRAIN_WGRAD=0: the split-K batched vendor GEMMs of rows_linear.py (A/B knob)
            gw = gb = None
            if WGRAD_X3 and need[name][0] and _capi.rows_wgrad_x3_supported(gy, x):
                gw, gb = _capi.rows_wgrad_x3(gy, x, bias=need[name][1], addend=addend)
            else:
                if addend is not None:
                    x = (x.view(-1, *addend.shape) + addend).view(x.shape)

                if need[name][0]:
                    gw = _RL.weight_grad(gy, x)
                if need[name][1]:
                    gb = _RL.bias_grad(gy)
            if gw is not None and post is not None:
                gw = post(gw)
            G[name] = [gw, gb]

        def ln_bwd(name, x, gy, w):
            gx, gw, gb = _capi.layernorm_bwd(x, gy, w, layer.norms[int(name[1])].eps)
            G[name] = [gw if need[name][0] else None, gb if need[name][1] else None]
            return gx

        g_y2 = g_y2.contiguous().view(R, E)
        q2, y0_2, s2 = q.view(R, E), y0.view(R, E), slots.view(R, E)
        # ================================================================ FFN block (recompute y1, hidden, pre-norm sum)
        # (every residual add, the ReLU's threshold_backward and the gradient sums ride in the GEMMs' store epilogues:
        #  fbbev_rows_linear_x3_train)
        co, c1, c2 = _frag(layer, 'co', *p['co']), _frag(layer, 'f1', *p['f1']), _frag(layer, 'f2', *p['f2'])
        x1 = _lin(s2, co, res=y0_2)                                                      # output_proj(slots) + residual
        y1 = _capi.layernorm(x1, p['n1'][0], p['n1'][1], layer.norms[1].eps)
        h = _lin(y1, c1, relu=True)
        x2 = _lin(h, c2, res=y1)
        g_x2 = ln_bwd('n2', x2, g_y2, p['n2'][0])
        del x2
        g_h = _lin(g_x2, _frag(layer, 'f2t', p['f2'][0], None, _t), mask=h)              # (g_x2 W2) * [h > 0]
        wgrad('f2', g_x2, h)
        del h
        g_y1 = _lin(g_h, _frag(layer, 'f1t', p['f1'][0], None, _t), res=g_x2)
        wgrad('f1', g_h, y1)
        del g_h, g_x2, y1
        # ================================================================ cross-attention tail
        g_x1 = ln_bwd('n1', x1, g_y1, p['n1'][0])
        del x1, g_y1
        g_slots = _lin(g_x1, _frag(layer, 'cot', p['co'][0], None, _t))
        wgrad('co', g_x1, s2)
        # ================================================================ depth-aware deformable cross-attention
        L, P = da.num_levels, da.num_points
        BN, S, _ = rows.shape
        HS = (Dh + 3) // 4 * 4
        hw = geo['hw']
        perm = _so_perm(da, M, L, P, dev)
        inv = _so_perm_inv(da, perm)
        cso = _frag(layer, 'cso_hm', p['cso'][0], p['cso'][1], lambda w_, b_: (w_[perm], b_[perm]))
        caw = _frag(layer, 'caw', *p['caw'])
        so = _lin(y0_2, cso, addend=pos)                                                  # (R, L*P*M*2) head-minor offsets
        aw = _softmax(_lin(y0_2, caw, addend=pos), L * P).view(R, M, L * P)
        from .backward_projection import _pad_interleave_rows
        cv = _frag(layer, 'cv_rows', p['cv'][0], p['cv'][1], lambda w_, b_: _pad_interleave_rows(w_, b_, M, Dh, HS, True))
        rows2 = rows.reshape(BN * S, E)
        g_d = torch.zeros_like(depth)
        Za = geo['mask'].shape[3]
        if DA_BWD_PLANES and _capi.da_cross_attn_bwd_planes_supported(B, BN // B, S, M, Dh, L, Q, P, Za, HS, hw, geo['bev_w']):
            # the forward's own head planes go straight to the gradient kernels, which WRITE the three large outputs in full: no row copy
            # of the camera tokens, no 0.5 GB of zero fills (fbbev_da_cross_attn_bwd_planes)
            g_v = torch.empty((BN, S, M, HS), dtype=torch.float32, device=dev)
            g_so, g_aw = torch.empty_like(so), torch.empty_like(aw)
            _capi.da_cross_attn_bwd_planes(planes_c, geo['ss'], geo['ls'], depth, geo['ref_cam'], geo['mask'], geo['qdepth'],
                                           so.view(B, Q, L, P, M, 2), aw.view(B, Q, M, L, P), g_slots.view(B, Q, E), ca.dbound[0],
                                           ca.dbound[2], 1 | 4, HS, g_v, g_d, g_so.view(B, Q, L, P, M, 2), g_aw.view(B, Q, M, L, P), hw,
                                           geo['bev_w'])
        else:
            v = _lin(rows2, cv).view(BN, S, M, HS)
            g_v = torch.zeros_like(v)
            g_so, g_aw = torch.zeros_like(so), torch.zeros_like(aw)
            _capi.da_cross_attn_bwd(v, geo['ss'], geo['ls'], depth, geo['ref_cam'], geo['mask'], geo['qdepth'],
                                    so.view(B, Q, L, P, M, 2), aw.view(B, Q, M, L, P), g_slots.view(B, Q, E), ca.dbound[0], ca.dbound[2],
                                    1 | 4, g_v, g_d, g_so.view(B, Q, L, P, M, 2), g_aw.view(B, Q, M, L, P), head_dim=Dh, level_hw=hw,
                                    bev_w=geo['bev_w'])
            del v
        del so, g_slots
        g_lg = _softmax_bwd(aw, g_aw, L * P).view(R, M * L * P)
        del g_aw, aw
        g_qp_c = _lin(g_so, _frag(layer, 'cso_hm_t', p['cso'][0], None, lambda w_, b_: (w_[perm].t().contiguous(), None)))
        _lin(g_lg, _frag(layer, 'cawt', p['caw'][0], None, _t), res=g_qp_c, out=g_qp_c)   # d / d (y0 + pos), both heads
        wgrad('cso', g_so, y0_2, post=lambda gw: gw[inv], addend=pos)
        if G['cso'][1] is not None:
            G['cso'][1] = G['cso'][1][inv]
        wgrad('caw', g_lg, y0_2, addend=pos)
        del g_so, g_lg
        g_y0 = g_x1 + g_qp_c                                                               # residual branch + the projections' input
        del g_x1
        g_rows = None
        g_v2 = g_v.view(BN * S, M * HS)
        if ctx.needs_input_grad[4]:
            g_rows = _lin(g_v2, _frag(layer, 'cv_rows_t', p['cv'][0], p['cv'][1],
                                      lambda w_, b_: (_pad_interleave_rows(w_, b_, M, Dh, HS, True)[0].t().contiguous(), None))).view(BN, S, E)

        def unpad_w(gw):                                                                 # (M*HS, E) chunk-major padded rows -> (M*Dh, E)
            return gw.view(HS // 4, M, 4, E).permute(1, 0, 2, 3).reshape(M, HS, E)[:, :Dh].reshape(M * Dh, E)
        wgrad('cv', g_v2, rows2, post=unpad_w)
        if G['cv'][1] is not None:
            G['cv'][1] = G['cv'][1].view(HS // 4, M, 4).permute(1, 0, 2).reshape(M, HS)[:, :Dh].reshape(M * Dh)
        del g_v, g_v2
        # ================================================================ self-attention tail (recompute the attention output)
        cso, caw = _frag(layer, 'sso', *p['sso']), _frag(layer, 'saw', *p['saw'])
        a = torch.empty_like(q)
        _capi.msda_self_fused(planes_s, geo['ref2d'], q, pos, cso.frag, cso.b, caw.frag, caw.b, sa.num_points, geo['bev_w'],
                              (geo['bev_h'], geo['bev_w']), a)
        a2 = a.view(R, E)
        x0 = _lin(a2, _frag(layer, 'so', *p['so']), res=q2)
        g_x0 = ln_bwd('n0', x0, g_y0, p['n0'][0])
        del x0, g_y0
        g_a = _lin(g_x0, _frag(layer, 'sot', p['so'][0], None, _t))
        wgrad('so', g_x0, a2)
        del a, a2
        # ================================================================ BEV self-attention (mmcv MultiScaleDeformableAttention)
        # sampling locations = reference point + offsets / (W, H) (multi_scale_deform_attn: offset_normalizer): the division lives in
        # a derived weight matrix, the reference point arrives as the GEMM's residual -- no element-wise pass; the offsets' gradient is
        # then the locations' gradient, and the weight gradient is rescaled row-wise at the end
        Ps = sa.num_points
        bh, bw = geo['bev_h'], geo['bev_w']
        nrm = _offset_normalizer(layer, M, Ps, bw, bh, dev)                                # (M*Ps*2): (W, H, W, H, ...)
        csn = _frag(layer, 'sso_n', p['sso'][0], p['sso'][1], lambda w_, b_: (w_ / nrm[:, None], b_ / nrm))
        loc = _lin(q2, csn, addend=pos, res=_ref_rows(layer, geo['ref2d'], M * Ps)).view(B, Q, M, 1, Ps, 2)
        aw = _softmax(_lin(q2, caw, addend=pos), Ps).view(R, M, Ps)
        v = _lin(q2, _frag(layer, 'sv', *p['sv'])).view(B, Q, M, Dh)
        g_v = torch.empty_like(v)
        g_loc, g_aw = torch.zeros_like(loc), torch.zeros_like(aw)
        _capi.msda_bwd(v, geo['ss_self'], geo['ls_self'], loc, aw.view(B, Q, M, 1, Ps), g_a.view(B, Q, E), g_v, g_loc,
                       g_aw.view(B, Q, M, 1, Ps), level_hw=[(bh, bw)])
        del v, loc, g_a
        g_loc = g_loc.view(R, M * Ps * 2)
        g_lg = _softmax_bwd(aw, g_aw, Ps).view(R, M * Ps)
        del g_aw, aw
        g_qp_s = _lin(g_loc, _frag(layer, 'sso_n_t', p['sso'][0], None, lambda w_, b_: ((w_ / nrm[:, None]).t().contiguous(), None)))
        _lin(g_lg, _frag(layer, 'sawt', p['saw'][0], None, _t), res=g_qp_s, out=g_qp_s)
        wgrad('sso', g_loc, q2, post=lambda gw: gw / nrm[:, None], addend=pos)
        if G['sso'][1] is not None:
            G['sso'][1] = G['sso'][1] / nrm
        wgrad('saw', g_lg, q2, addend=pos)
        del g_loc, g_lg
        g_pos = _capi.sum_leading(g_qp_c.view(B, Q, E), g_qp_s.view(B, Q, E)) if ctx.needs_input_grad[3] else None
        del g_qp_c
        g_v2 = g_v.view(R, E)
        g_q = _lin(g_v2, _frag(layer, 'svt', p['sv'][0], None, _t), res=g_x0)
        g_q.add_(g_qp_s)
        wgrad('sv', g_v2, q2)
        flat = []
        for n in NAMES:
            flat += G[n]
        return (None, None, g_q.view(B, Q, E) if ctx.needs_input_grad[2] else None, g_pos if ctx.needs_input_grad[3] else None,
                g_rows, g_d if ctx.needs_input_grad[5] else None, *flat)


def _softmax(x, group):
    """softmax over groups of `group` logits, in place (fbbev_softmax_groups); ATen for group sizes the kernel does not take"""
    if group in (4, 8, 16, 32):
        return _capi.softmax_groups(x, group, out=x)
    return x.view(-1, group).softmax(-1).view(x.shape)


def _softmax_bwd(y, gy, group):
    if group in (4, 8, 16, 32):
        return _capi.softmax_groups_bwd(y.reshape(-1), gy.reshape(-1), group, out=gy.reshape(-1))
    return torch._softmax_backward_data(gy.view(-1, group), y.view(-1, group), -1, torch.float32)


def _pos_table(bev_pos, Q, E):
    """the (Q, E) positional table behind `bev_pos`.  The encoder hands over a (bs, Q, E) stride-0 expand of it
    (positional_encoding.py:57-60); indexing that view would make autograd materialise a zero (bs, Q, E) gradient, copy one sample
    into it and sum the batch away again (0.3 ms at 160 000 queries) -- the view's base IS the table, taken when the layouts agree"""
    if bev_pos.dim() == 2:
        return bev_pos
    base = bev_pos._base
    if (base is not None and bev_pos.stride(0) == 0 and bev_pos.stride(1) == E and bev_pos.stride(2) == 1 and base.numel() == Q * E and
            base.is_contiguous() and base.data_ptr() == bev_pos.data_ptr() and base.shape[-1] == E):
        return base.view(Q, E)
    return bev_pos[0]


def _so_perm_inv(da, perm):
    if getattr(da, '_train_perm_inv_of', None) is not perm:
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        da._train_perm_inv, da._train_perm_inv_of = inv, perm
    return da._train_perm_inv


def _offset_normalizer(layer, M, Ps, bw, bh, dev):
    key = (M, Ps, bw, bh, str(dev))
    if getattr(layer, '_train_nrm_key', None) != key:
        layer._train_nrm = torch.tensor([float(bw), float(bh)], dtype=torch.float32).repeat(M * Ps).to(dev)
        layer._train_nrm_key = key
    return layer._train_nrm


def _ref_rows(layer, ref2d, reps):
    """(B, Q, 1, 2) reference points -> (B*Q, reps*2) rows (x, y, x, y, ...): the additive part of every sampling location of a query"""
    key = (ref2d.data_ptr(), ref2d._version, tuple(ref2d.shape), reps)
    if getattr(layer, '_train_ref_rows_key', None) != key:
        B, Q = ref2d.shape[:2]
        layer._train_ref_rows = ref2d.reshape(B * Q, 1, 2).expand(B * Q, reps, 2).reshape(B * Q, reps * 2).contiguous()
        layer._train_ref_rows_key = key
    return layer._train_ref_rows


def _so_perm(da, M, L, P, dev):
    """row permutation of sampling_offsets that makes the offsets head-minor, (L, P, M, 2) per query (DA_MSDeformableAttention.project_head_minor)"""
    key = (M, L, P, str(dev))
    if getattr(da, '_train_perm_key', None) != key:
        o = torch.arange(M * L * P).view(M, L, P).permute(1, 2, 0).reshape(-1)
        da._train_perm = (o[:, None] * 2 + torch.arange(2)[None]).reshape(-1).to(dev)
        da._train_perm_key = key
    return da._train_perm


def run_layer(layer, query, bev_pos, value_rows, pred_depth, ref_2d, ref_cam, mask, qdepth, spatial_shapes, level_start_index, bev_h, bev_w):
    """query (B, Q, E) rows, bev_pos (B, Q, E) stride-0 expand of the (Q, E) positional table, value_rows (B*Ncam, S, E) camera tokens,
    pred_depth (B, Ncam, DC, H0, W0) -> the layer's output rows"""
    from . import backward_projection as BP
    B, Q, E = query.shape
    dev = query.device
    hw = [(int(h), int(w)) for h, w in BP.host_values(spatial_shapes)]
    DC, H0, W0 = pred_depth.shape[2:]
    rk = (ref_2d.data_ptr(), ref_2d._version, tuple(ref_2d.shape), B, str(dev))
    if getattr(layer, '_train_ref_key', None) != rk:
        layer._train_ref, layer._train_ref_key = ref_2d.expand(B, Q, 1, 2).contiguous(), rk
    geo = dict(ref2d=layer._train_ref, ref_cam=ref_cam.contiguous().float(), mask=mask.contiguous(),
               qdepth=qdepth.squeeze(-1).contiguous().float(), ss=spatial_shapes.to(torch.int64).contiguous(),
               ls=level_start_index.to(torch.int64).contiguous(), hw=hw, min_w=min(w for _, w in hw), bev_h=bev_h, bev_w=bev_w,
               ss_self=BP.const_tensor([[bev_h, bev_w]], dev), ls_self=BP.const_tensor([0], dev))
    pos = _pos_table(bev_pos, Q, E)
    depth4 = pred_depth.reshape(-1, DC, H0, W0)
    return EncoderLayerFn.apply(layer, geo, query, pos, value_rows, depth4, *layer_params(layer))
