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
    for tv, cpl8, csplit, st in itertools.product((64, 128, 256), (0, 1), (1, 2), (0, 1, 2)):
        flags = _capi.pool_flags(store=st, cpl8=bool(cpl8), csplit=csplit)
        try:
            _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
            f = lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,  # noqa
                                                    idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, flags)
            ms = timeit(f)
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            rec = {'variant': f'tv{tv}_cpl{8 if cpl8 else 4}_cs{csplit}_st{st}', 'ms': ms, 'GBps_algo': algo / ms / 1e6,
                   'frac_8TBs': algo / ms / 1e6 / 8000, 'bits_equal_ref': same}
            if st == 0 and cpl8 == 1 and csplit == 1:   # pure pattern write with the same tiling
                _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, zero_counts, idx.n, B, Z, Y, X, ws, tv)
                rec['empty_index_ms'] = timeit(f)
                rec['empty_index_GBps'] = out.numel() * 4 / rec['empty_index_ms'] / 1e6
            print(json.dumps(rec), flush=True)
        except Exception as e:  # noqa
            print(json.dumps({'variant': f'tv{tv}_cpl{cpl8}_cs{csplit}_st{st}', 'error': str(e)}), flush=True)


if __name__ == '__main__':
    main()
