#!/usr/bin/env python3
"""Diagnose the dense pooling kernel inside the real pipeline (cold caches, exposed tail) vs warm
back-to-back launches.  Per-launch HIP-event timing in every mode."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S  # noqa: E402
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D  # noqa: E402


def per_launch(pre, f, iters=12, warm=3):
    evs = []
    for i in range(iters + warm):
        if pre is not None:
            pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record()
        if pre is None:
            torch.cuda.synchronize()
        if i >= warm:
            evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'BL2'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda:0')
    cfg = S.CONFIGS[name]
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    depth, ctx = depth.to(dev), ctx.to(dev)
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    out = torch.empty((B, C, Z, Y, X), device=dev)
    ws = vt._tile_ws(dev, B)
    state = {}

    def pre(tv, clamp=None, empty=False):
        idx = vt.build_index(vt.get_lidar_coor(*cam))
        state['feat'] = ctx.permute(0, 1, 3, 4, 2).contiguous()
        if clamp is not None:
            idx.interval_lengths.clamp_(max=clamp)
        counts = torch.zeros_like(idx.counts) if empty else idx.counts
        _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, counts, idx.n, B, Z, Y, X, ws, tv)
        state['idx'] = idx

    def run(tv, flags):
        idx = state['idx']
        _capi.bev_pool_v2_dense_fwd(depth, state['feat'], idx.ranks_depth, idx.ranks_feat, idx.interval_rank,
                                    idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, flags)

    idx0 = vt.build_index(vt.get_lidar_coor(*cam))
    P, I = idx0.counts.tolist()
    ln = idx0.interval_lengths[:I]
    print(json.dumps({'config': name, 'B': B, 'P': P, 'I': I, 'len_max': int(ln.max()), 'len_p999': float(torch.quantile(ln.float(), 0.999))}))
    variants = []
    for tv, cs, wg in ((256, 5, 256), (256, 10, 256), (512, 10, 256), (512, 20, 256), (1024, 10, 256), (1024, 20, 256),
                       (1024, 20, 128), (512, 5, 256), (128, 2, 256), (128, 1, 256)):
        for st in (4, 0):
            for lg in (None, 2, 4):
                variants.append((tv, cs, wg, lg, st))
    for tv, cs, wg, lg, st in variants:
        flags = _capi.pool_flags(store=st, csplit=cs, wg=wg, swizzle=lg is not None, swz_log2=lg or 0,
                                 cpl8=(C // cs) % 8 == 0)
        rec = {'variant': f'tv{tv}_cs{cs}_wg{wg}_swz{lg}_st{st}'}
        try:
            pre(tv)
            if st == 4:
                rec['warm_isolated_ms'] = round(per_launch(None, lambda: run(tv, flags)), 4)
                rec['pipeline_ms'] = round(per_launch(lambda: pre(tv), lambda: run(tv, flags)), 4)
            rec['pipeline_empty_ms'] = round(per_launch(lambda: pre(tv, empty=True), lambda: run(tv, flags)), 4)
        except Exception as e:  # noqa
            rec['error'] = str(e)
        print(json.dumps(rec), flush=True)
    # memset references, per-launch, warm and after the pipeline
    rec = {'variant': 'torch_zero_'}
    rec['warm_isolated_ms'] = round(per_launch(None, lambda: out.zero_()), 4)
    rec['pipeline_ms'] = round(per_launch(lambda: pre(64), lambda: out.zero_()), 4)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
