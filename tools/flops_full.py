#!/usr/bin/env python3
"""Multiply-add work of the detector's convolution stacks at the shipped config (B=1), counted with forward hooks on
the real modules (CPU run; the view transformation in between is replaced by a random volume of the right shape).
    python tools/flops_full.py [out.json]
FLOP = 2 * MACs; per stack and the five largest layers."""
import json
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd.fbocc import FBOCC  # noqa: E402


def main():
    from fb_bev_amd import configs
    cfg = configs.model_block()
    cfg.pop('type')
    torch.manual_seed(0)
    m = FBOCC(**cfg, execution=dict(with_cp=False)).eval()
    rows = []

    def hook(name):
        def fn(mod, inp, out):
            if isinstance(mod, nn.ConvTranspose3d):
                macs = inp[0].numel() / mod.in_channels * mod.in_channels * mod.out_channels * 8 / (mod.groups)
            elif isinstance(mod, (nn.Conv2d, nn.Conv3d)):
                k = 1
                for v in mod.kernel_size:
                    k *= v
                macs = out.numel() * mod.in_channels * k / mod.groups
            else:
                macs = out.numel() * mod.in_features
            rows.append((name, type(mod).__name__, list(out.shape), 2.0 * macs))
        return fn
    for name, mod in m.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose3d, nn.Linear)):
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        x = m.image_encoder(torch.randn(1, 6, 3, 256, 704))
        m.depth_net(x, torch.randn(1, 6, 27))
        feats = m.bev_encoder(torch.randn(1, 80, 100, 100, 8))
        m.occupancy_head(feats)
    stacks = {}
    for name, _, _, f in rows:
        top = name.split('.')[0]
        stacks[top] = stacks.get(top, 0.0) + f
    out = {'GFLOP_per_stack': {k: round(v / 1e9, 1) for k, v in stacks.items()},
           'GFLOP_3d_stacks': round(sum(v for k, v in stacks.items() if k in ('img_bev_encoder_backbone', 'img_bev_encoder_neck',
                                                                               'occupancy_head')) / 1e9, 1),
           'largest_layers': [dict(name=n, type=t, out=o, GFLOP=round(f / 1e9, 1)) for n, t, o, f in
                              sorted(rows, key=lambda r: -r[3])[:8]]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
