#!/usr/bin/env python3
"""GPU experiment: dense bev_pool_v2 forward variants (tile size, channels/lane, channel split, store
policy) vs two write-only ceilings (torch fill, and the same kernel on an EMPTY index = pure
pattern write).  Prints one JSON line per variant; run on the GPU box."""
import itertools
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S  # noqa: E402
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


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
    idx = vt.build_index(vt.get_lidar_coor(*cam))
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    out = torch.empty((B, C, Z, Y, X), device=dev)
    ws = vt._tile_ws(dev, B)
    P, I = idx.counts.tolist()
    H, W = cfg.feat_hw
    algo = 4 * B * cfg.n_cams * cfg.D * H * W + 4 * B * cfg.n_cams * H * W * C + 4 * (3 * P + 2 * I) + out.numel() * 4
    ms = timeit(lambda: out.zero_())
    print(json.dumps({'variant': 'torch_zero_', 'ms': ms, 'GBps': out.numel() * 4 / ms / 1e6}))
    zero_counts = torch.zeros(2, dtype=torch.int32, device=dev)
    ref = None
    combos = []
    for tv in (64, 128, 256, 512, 1024):
        for csplit in (1, 2, 4, 5, 10, 20):
            if C % (4 * csplit):
                continue
            cc = C // csplit
            lds = (cc * (tv + 4) + 3 * tv + 1024) * 4
            if lds > 60 * 1024:
                continue
            cpl8 = cc % 8 == 0
            if 128 // (cc // (8 if cpl8 else 4)) < 1:
                continue
            for swz in (False, True):
                for wg in (128, 256):
                    combos.append((tv, csplit, cpl8, swz, wg))
    mode = sys.argv[3] if len(sys.argv) > 3 else 'both'
    for tv, csplit, cpl8, swz, wg in combos:
        name_v = f'tv{tv}_cs{csplit}_cpl{8 if cpl8 else 4}_wg{wg}' + ('_swz' if swz else '')
        rec = {'variant': name_v}
        try:
            for st, tag in ((1, 'nt'), (0, 'plain')):
                flags = _capi.pool_flags(store=st, cpl8=cpl8, csplit=csplit, wg=wg, swizzle=swz)
                f = lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,  # noqa
                                                        idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, flags)
                _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, zero_counts, idx.n, B, Z, Y, X, ws, tv)
                ms = timeit(f, iters=10, warm=2)
                rec[f'empty_{tag}_GBps'] = round(out.numel() * 4 / ms / 1e6)
                if st == 1 and mode == 'both':
                    _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
                    ms = timeit(f, iters=10, warm=2)
                    if ref is None:
                        ref = out.clone()
                    rec.update(ms=round(ms, 4), frac_8TBs=round(algo / ms / 1e6 / 8000, 4), bits_equal_ref=bool(torch.equal(out, ref)))
            print(json.dumps(rec), flush=True)
        except Exception as e:  # noqa
            rec['error'] = str(e)
            print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
