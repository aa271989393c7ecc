#!/usr/bin/env python3
"""Time the fused geometry + ranking build (fbbev_lift_rank_build): python tools/time_rank.py CONFIG BATCH
Prints median ms per call and a checksum of the index tensors (must not change between tuning variants)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
from sweep_pool import per_launch


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    dev = torch.device('cuda:0'); cfg = S.CONFIGS[name]
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    idx = vt.build_index_from_cams(*cam)
    ms = per_launch(lambda: vt.build_index_from_cams(*cam), iters=20, warm=3)
    P, I = idx.counts.tolist()
    w = torch.arange(1, P + 1, device=dev, dtype=torch.int64)
    chk = [int(((t[:P].long() * w) % 1000003).sum().item()) for t in (idx.ranks_bev, idx.ranks_depth, idx.ranks_feat)]
    chk += [int(((t[:I].long() * w[:I]) % 1000003).sum().item()) for t in (idx.interval_starts, idx.interval_lengths)]
    print(json.dumps({'lib': os.path.basename(_capi.LIB_PATH), 'config': name, 'B': B, 'P': P, 'I': I, 'rank_build_ms': round(ms, 4), 'checksum': chk}))


if __name__ == '__main__':
    main()
