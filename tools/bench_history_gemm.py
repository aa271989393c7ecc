#!/usr/bin/env python3
"""Which library formulation runs the two 1x1x1 convs of the history fusion fastest?  (B=1, T+1=17, C=80, N=Z*Y*X)"""
import json, sys
import torch
dev = torch.device('cuda:0')
T1, C, N = 17, 80, int(sys.argv[1]) if len(sys.argv) > 1 else 80000
x = torch.randn(T1, C, N, device=dev)
w = torch.randn(C, C, device=dev) * 0.1
b = torch.randn(T1, C, 1, device=dev)
w2 = torch.randn(C, T1 * C, device=dev) * 0.05
b2 = torch.randn(1, C, 1, device=dev)


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


res = {}
res['time_baddbmm_expand'] = timed(lambda: torch.baddbmm(b, w.unsqueeze(0).expand(T1, C, C), x))
wc = w.unsqueeze(0).expand(T1, C, C).contiguous()
res['time_baddbmm_contig_w'] = timed(lambda: torch.baddbmm(b, wc, x))
res['time_matmul_bcast'] = timed(lambda: torch.matmul(w, x) + b)
res['time_conv3d'] = timed(lambda: torch.nn.functional.conv3d(x.view(T1, C, 8, 100, N // 800), w.view(C, C, 1, 1, 1)))
xt = x.permute(0, 2, 1).contiguous()      # (T1, N, C): token-major
res['time_linear_token_major'] = timed(lambda: torch.nn.functional.linear(xt, w))
y = torch.randn(1, T1 * C, N, device=dev)
res['cat_baddbmm'] = timed(lambda: torch.baddbmm(b2, w2.unsqueeze(0), y))
res['cat_addmm_2d'] = timed(lambda: torch.addmm(b2.view(C, 1), w2, y.view(T1 * C, N)))
res['cat_conv3d'] = timed(lambda: torch.nn.functional.conv3d(y.view(1, T1 * C, 8, 100, N // 800), w2.view(C, T1 * C, 1, 1, 1)))
res['relu_inplace_time_tensor'] = timed(lambda: x.relu_())
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
