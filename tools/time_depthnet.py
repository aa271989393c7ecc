#!/usr/bin/env python3
"""CM_DepthNet at the shipped FB-OCC shapes (256 -> 512 channels, 16x44 feature map, 6 cameras) on one GPU:
fp32 NCHW (the reference's setup) vs channels-last vs channels-last + bf16 autocast.  python tools/time_depthnet.py [B]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd.depth_net import CM_DepthNet


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(B, 6, 256, 16, 44, device=dev)
    mlp = torch.randn(B, 6, 27, device=dev)
    res = {'B': B}
    ref_out = None
    only = sys.argv[2] if len(sys.argv) > 2 else None          # e.g. bf16_channels_last: one variant (counter passes)
    for tag, kw in (('fp32_nchw', dict(channels_last=False)), ('fp32_channels_last', dict(channels_last=True)),
                    ('bf16_channels_last', dict(channels_last=True, compute_dtype=torch.bfloat16))):
        if only and tag != only:
            continue
        torch.manual_seed(1)
        net = CM_DepthNet(in_channels=256, context_channels=80, depth_channels=80, downsample=16, use_dcn=False,
                          grid_config={'depth': [2.0, 42.0, 0.5]}, **kw).to(dev).eval()
        with torch.no_grad():
            ms = timed(lambda: net(x, mlp))
            c, d = net(x, mlp)
        if ref_out is None:
            ref_out = (c, d)
        res[tag + '_ms'] = round(ms, 3)
        res[tag + '_max_abs_diff_depth'] = float((d - ref_out[1]).abs().max())
    # ~ flops: reduce 3x3 256->512, 6 3x3 512->512 (blocks), ASPP (1x1 + three 3x3 512->512 + 1x1 2560->512), heads
    px = B * 6 * 16 * 44
    flops = 2 * px * (9 * 256 * 512 + 6 * 9 * 512 * 512 + 512 * 512 + 3 * 9 * 512 * 512 + 2560 * 512 + 512 * 80 * 2 + 4 * 512 * 512)
    res['approx_GFLOP'] = round(flops / 1e9, 1)
    if 'bf16_channels_last_ms' in res:
        res['bf16_TFLOPs'] = round(flops / res['bf16_channels_last_ms'] / 1e9, 1)
    if 'fp32_channels_last_ms' in res:
        res['fp32_TFLOPs'] = round(flops / res['fp32_channels_last_ms'] / 1e9, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
