#!/usr/bin/env python3
"""Per-launch timing sweep of the dense bev_pool_v2 kernel: python tools/sweep_pool.py CONFIG BATCH [f32|bf16|f16]"""
import itertools, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D


def per_launch(f, iters=10, warm=2):
    ts = []
    for i in range(iters + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    storage = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    odt = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[storage]
    esz = 4 if storage == 'f32' else 2
    dev = torch.device('cuda:0'); cfg = S.CONFIGS[name]
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(cfg, B, seed=0); depth, ctx = depth.to(dev), ctx.to(dev)
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    Z, Y, X = vt.grid_zyx; C = cfg.channels
    idx = vt.build_index_from_cams(*cam); feat = _capi.nchw_to_nhwc(ctx)
    out = torch.empty((B, C, Z, Y, X), device=dev, dtype=odt)
    ws = torch.empty(_capi.pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8, device=dev)
    P, I = idx.counts.tolist(); H, W = cfg.feat_hw
    algo = 4 * B * cfg.n_cams * cfg.D * H * W + 4 * B * cfg.n_cams * H * W * C + 4 * (3 * P + 2 * I) + out.numel() * esz
    print(json.dumps({'config': name, 'B': B, 'storage': storage, 'P': P, 'I': I, 'algo_bytes': algo}))
    ref = None
    tvs = (32, 64, 128) if storage == 'f32' else (64, 128, 256)
    for tv, cs, wg, lg in itertools.product(tvs, (1, 2), (256, 128), (4, 5, 6)):
        if C % (4 * cs):
            continue
        cc = C // cs
        if (cc * (tv + 4) * esz // 4 + 3 * tv + 1024) * 4 > 64 * 1024:
            continue
        for cpl8, deep in itertools.product((True, False), (0,)):
            if cpl8 and cc % 8:
                continue
            if wg // (cc // (8 if cpl8 else 4)) < 1:
                continue
            flags = _capi.pool_flags(store=4, csplit=cs, wg=wg, swizzle=lg is not None, swz_log2=lg or 0, cpl8=cpl8) | deep | \
                {'f32': 0, 'bf16': 0x800000, 'f16': 0x1000000}[storage]
            _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
            f = lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                                    idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, flags)
            ms = per_launch(f)
            if ref is None:
                ref = out.clone()
            print(json.dumps({'variant': f'tv{tv}_cs{cs}_wg{wg}_swz{lg}_cpl{8 if cpl8 else 4}' + ('_deep' if deep else ''), 'ms': round(ms, 4),
                              'frac': round(algo / ms / 1e6 / 8000, 3), 'bits_ok': bool(torch.equal(out, ref))}), flush=True)


if __name__ == '__main__':
    main()
