#!/usr/bin/env python3
"""HIP-event times of the row kernels of the training path at BASELINE configs[2] sizes (R = 160 000 rows): fbbev_rows_wgrad_x3 per
layer shape with / without the periodic addend, fbbev_rows_linear_x3[_train] per shape.  python tools/time_rows_kernels.py -> JSON lines"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


def main():
    dev = torch.device('cuda:0')
    R, Q = 160000, 40000
    g = torch.Generator(device='cpu').manual_seed(0)
    pos = torch.randn(Q, 80, generator=g).to(dev)
    for O, I in ((512, 80), (256, 80), (320, 80), (80, 80), (64, 80), (32, 80), (80, 320), (96, 80)):
        gy, x = torch.randn(R, O, generator=g).to(dev), torch.randn(R, I, generator=g).to(dev)
        rec = {'kernel': 'rows_wgrad_x3', 'O': O, 'I': I, 'R': R, 'MB': (gy.numel() + x.numel()) * 4 / 1e6,
               'us': timed(lambda: _capi.rows_wgrad_x3(gy, x)), 'us_no_bias': timed(lambda: _capi.rows_wgrad_x3(gy, x, bias=False))}
        if I == 80:
            rec['us_addend'] = timed(lambda: _capi.rows_wgrad_x3(gy, x, addend=pos))
        rec['TBps'] = rec['MB'] / rec['us']
        print(json.dumps(rec), flush=True)
    for I, O in ((80, 512), (80, 256), (80, 320), (80, 80), (512, 80), (256, 80), (320, 80), (80, 64), (64, 80)):
        x, w = torch.randn(R, I, generator=g).to(dev), (torch.randn(O, I, generator=g) * 0.1).to(dev)
        frag = _capi.rows_linear_x3_fragments(w)
        res = torch.randn(R, O, generator=g).to(dev)
        out = torch.empty(R, O, device=dev)
        rec = {'kernel': 'rows_linear_x3', 'I': I, 'O': O, 'MB': (x.numel() + out.numel()) * 4 / 1e6,
               'us': timed(lambda: _capi.rows_linear_x3(x, frag, None, O, out=out)),
               'us_train_res': timed(lambda: _capi.rows_linear_x3_train(x, frag, None, O, residual=res, out=out)),
               'us_train_mask': timed(lambda: _capi.rows_linear_x3_train(x, frag, None, O, mask=res, out=out))}
        rec['TBps'] = rec['MB'] / rec['us']
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
