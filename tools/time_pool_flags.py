#!/usr/bin/env python3
"""Per-launch time of the dense bev_pool_v2 kernel for a list of (tile_voxels, flag word) variants:
    python tools/time_pool_flags.py CONFIG BATCH [storage] [tv:flags ...]     (default: the module's tiling + the other tile sizes)
Used in round 2 for the early-zero-store experiment (profiles/r02_exp_pool_early_zero_stores.jsonl: 3.6x SLOWER -- the
stores issued before the gathers hold the in-order vmcnt queue until the write stream has acknowledged them; removed)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
from sweep_pool import per_launch


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    storage = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    dev = torch.device('cuda:0'); cfg = S.CONFIGS[name]
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(cfg, B, seed=0); depth, ctx = depth.to(dev), ctx.to(dev)
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    Z, Y, X = vt.grid_zyx; C = cfg.channels
    idx = vt.build_index_from_cams(*cam); feat = _capi.nchw_to_nhwc(ctx)
    tv0, fl0 = vt.tiling(cfg.n_cams)
    dt = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[storage]
    esz = 4 if storage == 'f32' else 2
    out = torch.empty((B, C, Z, Y, X), device=dev, dtype=dt)
    P, I = idx.counts.tolist(); H, W = cfg.feat_hw
    algo = 4 * B * cfg.n_cams * cfg.D * H * W + 4 * B * cfg.n_cams * H * W * C + 4 * (3 * P + 2 * I) + out.numel() * esz
    ref = None
    extra = [a for a in sys.argv[4:] if ':' in a]
    variants = [(tv0, fl0)] + [(int(a.split(':')[0]), int(a.split(':')[1], 0)) for a in extra]
    if not extra:
        variants += [(tv, fl) for tv, fl in ((64, _capi.pool_flags(csplit=1)), (128, _capi.DEFAULT_POOL_FLAGS), (256, _capi.DEFAULT_POOL_FLAGS))
                     if (tv, fl) != (tv0, fl0)]
    for tv, fl in variants:
        ws = torch.empty(_capi.pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8, device=dev)
        _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
        f = lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                                idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, fl)
        try:
            ms = per_launch(f, iters=20, warm=3)
        except Exception as e:      # unsupported combination
            print(json.dumps({'config': name, 'B': B, 'tv': tv, 'flags': hex(fl), 'error': str(e)[:80]})); continue
        if ref is None:
            ref = out.clone()
        print(json.dumps({'config': name, 'B': B, 'storage': storage, 'tv': tv, 'flags': hex(fl),
                          'ms': round(ms, 4), 'frac_of_8TBs': round(algo / ms / 1e6 / 8000, 3), 'bits_equal_first': bool(torch.equal(out, ref))}), flush=True)


if __name__ == '__main__':
    main()
