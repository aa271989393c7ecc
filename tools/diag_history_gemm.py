#!/usr/bin/env python3
"""Which of the two conv formulations is the accurate one at full size?  fp64 GPU einsum as the referee."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, T, n = 80, 16, 80000
x = torch.randn(T + 1, C, n, device=dev)
w = (torch.rand(C, C, device=dev) - 0.5) * 0.2
b = torch.randn(C, device=dev) * 0.1
ref64 = torch.einsum('oc,tcn->ton', w.double(), x.double()) + b.double()[None, :, None]
conv = torch.nn.functional.conv3d(x.view(T + 1, C, 8, 100, 100), w.view(C, C, 1, 1, 1), b).view(T + 1, C, n)
bad = torch.baddbmm(b.view(1, C, 1), w.unsqueeze(0).expand(T + 1, C, C), x)
mm = torch.matmul(w, x) + b[None, :, None]
print(json.dumps({'conv3d_vs_fp64': (conv.double() - ref64).abs().max().item(), 'baddbmm_vs_fp64': (bad.double() - ref64).abs().max().item(),
                  'matmul_vs_fp64': (mm.double() - ref64).abs().max().item(), 'scale': ref64.abs().max().item()}))
x2 = torch.randn(1, (T + 1) * C, n, device=dev)
w2 = (torch.rand(C, (T + 1) * C, device=dev) - 0.5) * 0.05
ref64 = torch.einsum('oc,bcn->bon', w2.double(), x2.double())
conv = torch.nn.functional.conv3d(x2.view(1, -1, 8, 100, 100), w2.view(C, -1, 1, 1, 1)).view(1, C, n)
bad = torch.baddbmm(torch.zeros(1, C, 1, device=dev), w2.unsqueeze(0), x2)
print(json.dumps({'K1360_conv3d_vs_fp64': (conv.double() - ref64).abs().max().item(), 'K1360_baddbmm_vs_fp64': (bad.double() - ref64).abs().max().item(),
                  'scale': ref64.abs().max().item()}))
