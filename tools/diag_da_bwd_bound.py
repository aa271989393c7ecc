#!/usr/bin/env python3
"""Per-gradient error of fbbev_da_cross_attn_bwd / _bwd_ws against the fp64 autograd of the oracle at large Q (diagnostic of
tests/test_gpu_backward_projection.py::test_fused_backward_kernels_within_a_bound_of_fp64_autograd)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fb_bev_amd import _capi
from da_cases import da_case
dev = torch.device('cuda:0')
cases = [(9, dict(B=1, Q=10000, E=80, M=8, shapes=((16, 44), (32, 88), (8, 22), (4, 11)), DC=20)),
         (10, dict(B=1, Q=40000, E=80, M=8, shapes=((16, 44),), DC=20)),
         (11, dict(B=1, Q=2500, E=80, M=8, shapes=((16, 44),), DC=20))]
for seed, kw in cases:
    args, exp, leaves = da_case(seed, grad=True, **kw)
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    value, ss, ls, pred4, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    names = ['key', 'pred'] + [k for k in sorted(leaves['Pm']) if 'output_proj' not in k]
    wrt = [leaves['key'], leaves['pred']] + [leaves['Pm'][k] for k in sorted(leaves['Pm']) if 'output_proj' not in k]
    ref = torch.autograd.grad(exp, wrt, grad_outputs=g, retain_graph=True, allow_unused=True)
    Dh = value.shape[-1]; HS = (Dh + 3) // 4 * 4
    f32 = lambda t: t.detach().float().contiguous()
    vp = torch.zeros(value.shape[:-1] + (HS,)); vp[..., :Dh] = f32(value)
    t = lambda x: x.to(dev).contiguous()
    a = [t(vp), t(ss), t(ls), t(f32(pred4)), t(f32(ref_cam)), t(mask), t(f32(qdepth)), t(f32(offsets)), t(f32(attn)), t(f32(g)), d0, dstep, 0]
    for lds in (False, True):
        gv, gd, go, ga = (torch.zeros_like(x) for x in (a[0], a[3], a[7], a[8]))
        _capi.da_cross_attn_bwd(*a, gv, gd, go, ga, head_dim=Dh, lds_planes=lds, level_hw=[tuple(int(x) for x in hw) for hw in ss.tolist()])
        torch.cuda.synchronize()
        row = dict(seed=seed, Q=kw['Q'], lds_planes=lds)
        mines = torch.autograd.grad([value, pred4, offsets, attn], wrt,
                                    grad_outputs=[gv[..., :Dh].cpu().double(), gd.cpu().double(), go.cpu().double(), ga.cpu().double()],
                                    retain_graph=True, allow_unused=True)
        for name, mine, r in zip(names, mines, ref):
            if r is None:
                continue
            name = name.replace('a.deformable_attention.', '')
            d = (mine - r).abs()
            row[name] = dict(max_err=float(d.max()), scale=float(r.abs().max()), rel_to_scale=float(d.max() / r.abs().max()),
                             frac_over_1e4rel_5e5scale=float((d > 1e-4 * r.abs() + 5e-5 * r.abs().max()).double().mean()))
        print(json.dumps(row), flush=True)
