"""Forward-backward view transformation: the hot-path slice of FBOCC.extract_img_bev_feat
(mmdet3d/models/fbbev/detectors/fbocc.py:344-366): forward projection (lift-splat) -> backward
projection refinement of the Z-mean BEV -> re-add broadcast over Z.  Inputs are what the depth net
produces (`context`, `depth`) plus `cam_params = img_inputs[1:7]`."""
import torch
import torch.nn as nn

from . import _capi
from . import backward_projection as BP
from . import train_path as TP
from .view_transformer import LSSViewTransformerFunction3D


import os as _os
ZMEAN_ROWS = _os.environ.get('FBBEV_ZMEAN_ROWS', '1') != '0'     # A/B knob (round 6): inference, the Z-mean written as query rows + bev_embedding
_ONE_OP = _os.environ.get('FBBEV_LSS_ZMEAN', '1') != '0'      # A/B knob: volume + Z-mean as one differentiable op (training)


class _ZMean(torch.autograd.Function):
    """bev_feat.mean(-1) of the (B,C,Y,X,Z) view of a (B,C,Z,Y,X) volume (fbocc.py:359) as one HBM-bound pass."""

    @staticmethod
    def forward(ctx, vol):
        ctx.Z = vol.shape[-1]
        return _capi.volume_zreduce(vol, vol.shape[-1])

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.Z)[..., None].expand(*g.shape, ctx.Z)


class _ReAdd(torch.autograd.Function):
    """refined[..., None] + bev_feat (fbocc.py:365-366); the backward's Z-sum for `refined` is one HBM-bound pass instead of an
    ATen reduction over the strided last dimension (0.93 -> 0.2 ms at the configs[2] grid, B = 4)."""

    @staticmethod
    def forward(ctx, refined, vol):
        return refined[..., None] + vol

    @staticmethod
    def backward(ctx, g):
        if _capi.volume_zreduce_supported(g):          # the gradient has the volume's own memory layout
            return _capi.volume_zreduce(g, 1.0), g
        if _capi.volume_zlast_supported(g):            # ... or is contiguous in the output's (B,C,Y,X,Z) shape
            return _capi.volume_zreduce_inner(g, 1.0), g
        return g.sum(-1), g


class FBViewTransform(nn.Module):
    def __init__(self, forward_projection, backward_projection=None, readd=True):
        super().__init__()
        fp = dict(forward_projection)
        assert fp.pop('type') == 'LSSViewTransformerFunction3D'
        self.forward_projection = LSSViewTransformerFunction3D(**fp)
        self.backward_projection = BP.build(backward_projection) if backward_projection is not None else None
        self.readd = readd
        self.write_once = True      # inference: Z-mean from the index tensors + re-add in the pooling store (volume written once)

    def forward(self, cam_params, context, depth, img_metas=None, bev_mask=None, mlvl_feats=None):
        """mlvl_feats: optional list of (B,N,C,H_l,W_l) image features for the backward projection (default
        [context], as fbocc.py:357 passes); BASELINE configs[2] uses 4 levels."""
        fp = self.forward_projection
        feats = mlvl_feats if mlvl_feats is not None else [context]
        needs_grad = torch.is_grad_enabled() and (context.requires_grad or depth.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if (self.write_once and self.backward_projection is not None and self.readd and fp.fused and not fp.extra_relu and not needs_grad
                and context.is_cuda and fp._fused_supported(context.shape[2])):
            # inference: the volume is written ONCE.  The reference writes it (bev_pool_v2), reads it for the Z-mean
            # (fbocc.py:359) and reads + re-writes it for the re-add (:365-366); here the Z-mean comes straight from the
            # index tensors and the refined BEV is added in the store epilogue of the one dense pooling pass.
            # round 5: what the backward projection can do without the Z-mean (camera-token rows, their value planes, point sampling:
            # ~90 us at BASELINE configs[2]) starts on a side stream and runs under the ranking chain + Z-mean below, whose
            # latency-bound kernels leave most of the chip idle
            pre = self.backward_projection.prefetch(feats, cam_params) if hasattr(self.backward_projection, 'prefetch') else None
            parts = fp.pooling_inputs(cam_params, context, depth)
            kw = {} if pre is None else {'_pre': pre}
            bp = self.backward_projection
            rows = None
            if ZMEAN_ROWS and hasattr(bp, 'query_row_bias') and context.dtype == torch.float32:
                # round 6: the Z-mean leaves its kernel as the backward projection's query rows (+ bev_embedding): no transposing pass
                bias = bp.query_row_bias(context.shape[2], fp.grid_zyx)
                rows = fp.pooled_zmean_rows(parts, bias) if bias is not None else None
            if rows is not None:
                refined = bp(feats, img_metas, lss_rows=rows, cam_params=cam_params, bev_mask=bev_mask, gt_bboxes_3d=None,
                             pred_img_depth=depth, **kw)
            else:
                lss_mean = fp.pooled_zmean(parts)
                refined = bp(feats, img_metas, lss_bev=lss_mean, cam_params=cam_params, bev_mask=bev_mask,
                             gt_bboxes_3d=None, pred_img_depth=depth, **kw)
            return fp.pooled_volume(parts, addend=refined)
        if (self.write_once and self.backward_projection is not None and self.readd and needs_grad and TP.TRAIN_FUSED and
                TP.write_once_supported(fp, context) and depth.dtype == torch.float32):
            # training (round 6): the volume is written once here too -- Z-mean from the index tensors, the refined BEV added in the
            # pooling store, ONE pooling backward with the mean's gradient folded in (train_path.WriteOnce)
            parts = fp.pooling_inputs(cam_params, context, depth)
            shared = {'defer': True}
            lss_mean = TP.WriteOnce.ZMean.apply(context, depth, fp, parts, shared)
            refined = self.backward_projection(feats, img_metas, lss_bev=lss_mean, cam_params=cam_params, bev_mask=bev_mask,
                                               gt_bboxes_3d=None, pred_img_depth=depth)
            return TP.WriteOnce.PoolAdd.apply(context, depth, refined, fp, parts, shared)
        both = fp.forward_with_zmean(cam_params, context, depth) if (self.backward_projection is not None and needs_grad and _ONE_OP) else None
        if both is not None:
            # training: the volume and its Z-mean leave the lift-splat as ONE differentiable op -- the mean's gradient is folded
            # into the pooling backward instead of being expanded and added to the volume's gradient (0.35 ms at the configs[2] grid)
            bev_feat, lss_mean = both
            fast = True
        else:
            bev_feat = fp(cam_params, context, depth)                             # (B,C,Y,X,Z)   fbocc.py:344-345
            if self.backward_projection is None:
                return bev_feat
            fast = _capi.volume_zreduce_supported(bev_feat)                       # the volume exists; one pass per reduction
            lss_mean = _ZMean.apply(bev_feat) if fast else bev_feat.mean(-1)
        refined = self.backward_projection(feats, img_metas, lss_bev=lss_mean,
                                           cam_params=cam_params, bev_mask=bev_mask, gt_bboxes_3d=None, pred_img_depth=depth)   # :357-363
        if not self.readd:
            return refined
        return _ReAdd.apply(refined, bev_feat) if fast and refined.dtype == bev_feat.dtype else refined[..., None] + bev_feat   # :365-368
