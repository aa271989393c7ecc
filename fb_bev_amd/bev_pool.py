"""Host-side mirror of mmdet3d/ops/bev_pool_v2/bev_pool.py (QuickCumsumCuda :11-80,
bev_pool_v2 :83-89, TRTBEVPoolv2 :92-141): same names, argument meaning and autograd contract (gradients for `depth`
and `feat` only), running on the HIP kernels of libfbbev_hip.so.
"""
import torch

from . import bev_pool_v2_ext

__all__ = ['bev_pool_v2', 'QuickCumsumCuda', 'TRTBEVPoolv2', 'intervals_over']


def intervals_over(sorted_keys):
    """Run starts / lengths over a sorted 1-D key tensor (the construction of bev_pool.py:47-54
    and view_transformer.py:593-602).  Needs one host sync (nonzero) -- the fused path in
    view_transformer.LiftSplat avoids it; this exists for the reference-compatible entry."""
    n = sorted_keys.shape[0]
    head = torch.ones(n, dtype=torch.bool, device=sorted_keys.device)
    head[1:] = sorted_keys[1:] != sorted_keys[:-1]
    starts = head.nonzero().squeeze(1).int()
    lengths = torch.empty_like(starts)
    if starts.numel():
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = n - starts[-1]
    return starts, lengths


class QuickCumsumCuda(torch.autograd.Function):
    """bev_pool.py:11-80.  forward -> (B,Z,Y,X,C) with only non-empty voxels written."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        ranks_bev = ranks_bev.contiguous().int()
        ranks_depth = ranks_depth.contiguous().int()
        ranks_feat = ranks_feat.contiguous().int()
        interval_lengths = interval_lengths.contiguous().int()
        interval_starts = interval_starts.contiguous().int()
        out = feat.new_zeros(bev_feat_shape)
        bev_pool_v2_ext.bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                                            interval_lengths, interval_starts)
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        # bev_pool.py:44-54: regroup the points by feature pixel.  Stable sort => deterministic
        # summation order (the reference's argsort is unstable).
        ranks_feat, order = torch.sort(ranks_feat, stable=True)
        ranks_depth = ranks_depth[order].contiguous()
        ranks_bev = ranks_bev[order].contiguous()
        ranks_feat = ranks_feat.contiguous()
        starts_bp, lengths_bp = intervals_over(ranks_feat)
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        bev_pool_v2_ext.bev_pool_v2_backward(out_grad.contiguous(), depth_grad, feat_grad, depth, feat,
                                             ranks_depth, ranks_feat, ranks_bev, lengths_bp, starts_bp)
        return depth_grad, feat_grad, None, None, None, None, None, None


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
    """bev_pool.py:83-89 -> (B,C,Z,Y,X) contiguous."""
    x = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths)
    return x.permute(0, 4, 1, 2, 3).contiguous()


class TRTBEVPoolv2(torch.autograd.Function):
    """ONNX export stand-in of the op (ops/bev_pool_v2/bev_pool.py:92-141): `symbolic` emits the custom node
    `mmdeploy::bev_pool_v2` with the reference's attribute names; `forward` is the eager equivalent on the HIP op for a
    single-sample, Z = 1 grid -> (1, out_height, out_width, C)."""

    @staticmethod
    def symbolic(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out_height=128,
                 out_width=128):
        return g.op('mmdeploy::bev_pool_v2', depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                    interval_lengths, out_height_i=out_height, out_width_i=out_width)

    @staticmethod
    def forward(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out_height=128,
                out_width=128):
        n, d, h, w = depth.shape
        feat = feat.view(1, n, feat.shape[3], h, w).permute(0, 1, 3, 4, 2)
        depth = depth.view(1, n, d, h, w)
        bev_feat_shape = (1, 1, out_height, out_width, feat.shape[-1])                   # (B, Z, Y, X, C)
        bev_feat = bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                               interval_lengths)
        return bev_feat.squeeze(2).permute(0, 2, 3, 1)
