#!/usr/bin/env python3
"""per-tensor deviation of the one-node training route's gradients from the composite route's: python tools/dbg_train_grads.py CONFIG B LEVELS"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import train_path as T
from fb_bev_amd import train_path as TP
name, B, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda:0')
res = {}
for fused in (False, True):
    TP.TRAIN_FUSED = fused
    pc, m, cam, depth, ctx, mlvl = T.build(name, B, L, dev)
    step, leaves, gout = T.make_step(m, cam, depth, ctx, mlvl, dev, pc, B)
    out = step()
    names = [n for n, _ in m.named_parameters()] + ['depth', 'ctx'] + [f'mlvl{i}' for i in range(1, len(mlvl or []))]
    res[fused] = (out.detach(), {n: (None if t.grad is None else t.grad.detach().clone()) for n, t in zip(names, leaves)})
print('forward', (res[True][0] - res[False][0]).abs().max().item(), res[False][0].abs().max().item())
for n, a in res[True][1].items():
    b = res[False][1][n]
    if a is None or b is None:
        print(f'{n:90s} fused {"None" if a is None else "set"} composite {"None" if b is None else "set"}')
        continue
    s = b.abs().max().item()
    e = (a - b).abs()
    print(f'{n[-80:]:80s} scale {s:10.4g} max_err/scale {e.max().item() / (s + 1e-30):9.2e} mean_err/scale {e.mean().item() / (s + 1e-30):9.2e}')
