"""Time the sync-free fused backward (fbbev_bev_pool_v2_dense_bwd) against the re-sort path the reference's
autograd function takes (argsort ranks_feat + mask intervals + permuted gradient copy + grad kernel).
Usage: python tools/time_bwd.py [CONFIG] [B]   -> JSON lines"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import _capi, bev_pool_v2_ext, synthetic as S  # noqa: E402
from fb_bev_amd.bev_pool import intervals_over  # noqa: E402
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'BL2'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda:0')
    cfg = S.CONFIGS[name]
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    depth, ctx = depth.to(dev), ctx.to(dev)
    feat = _capi.nchw_to_nhwc(ctx)
    idx = vt.build_index_from_cams(*cam)
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    N, D, H, W = depth.shape[1:]
    og = torch.randn((B, C, Z, Y, X), device=dev)
    dg, fg = torch.empty_like(depth), torch.empty_like(feat)
    ws = torch.empty(_capi.pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X), dtype=torch.uint8, device=dev)

    def fused():
        _capi.bev_pool_v2_dense_bwd(og, depth, feat, idx.ranks_depth, idx.interval_rank, idx.interval_starts,
                                    idx.counts, idx.n, (Z, Y, X), dg, fg, ws)

    def resort():
        rb, rd, rf, _, _ = idx.exact()
        rf2, order = torch.sort(rf, stable=True)
        rd2, rb2 = rd[order].contiguous(), rb[order].contiguous()
        st, ln = intervals_over(rf2)
        ogl = og.permute(0, 2, 3, 4, 1).contiguous()
        dg2, fg2 = torch.zeros_like(depth), torch.zeros_like(feat)
        bev_pool_v2_ext.bev_pool_v2_backward(ogl, dg2, fg2, depth, feat, rd2, rf2.contiguous(), rb2, ln, st)
        return dg2, fg2

    t_f = timed(fused)
    t_r = timed(resort)
    dg2, fg2 = resort()
    P, I = idx.counts.tolist()
    alg = og.numel() * 4 + I * C * 4 * 2 + P * C * 4 + depth.numel() * 8 + feat.numel() * 8 + idx.n * 4
    print(json.dumps({'config': name, 'B': B, 'P': P, 'I': I, 'fused_bwd_ms': round(t_f, 4), 'resort_bwd_ms': round(t_r, 4),
                      'speedup': round(t_r / t_f, 2), 'fused_algorithmic_GB': round(alg / 1e9, 3),
                      'fused_GBps': round(alg / t_f / 1e6, 1),
                      'max_abs_diff_feat_grad': (fg - fg2).abs().max().item(),
                      'max_abs_diff_depth_grad': (dg - dg2).abs().max().item()}))


if __name__ == '__main__':
    main()
