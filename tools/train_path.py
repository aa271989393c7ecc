#!/usr/bin/env python3
"""Training step of the view-transformation PATH (forward + backward of FBViewTransform: lift-splat, Z-mean, backward projection,
re-add) on one GPU with the upstream gradient HANDED OVER in the output's own memory layout -- the protocol of bench.py's
`fb_projection_train` leg (BASELINE configs[2] / [3]; reference: bev_pool.py:40-80, bev_pool_cuda.cu:64-118,
multi_scale_deformable_attn_function.py:137-172).

python tools/train_path.py [CONFIG] [BATCH] [LEVELS] [--steps N] [--profile-steps N] [--sites]   -> one JSON line
With --profile-steps the script runs ONLY warm-up + N steps (what a `rocprofv3 --kernel-trace --stats` pass wants)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd import synthetic as S                               # noqa: E402


def build(name, B, levels, dev, seed=0, feat_grad=True):
    d = S.fb_path_step(name, B, levels, dev, seed=seed, feat_grad=feat_grad)
    build.last = d
    return d['pc'], d['model'], d['cam'], d['depth'], d['ctx'], d['mlvl']


def make_step(m, cam, depth, ctx, mlvl, dev, pc, B):
    d = build.last
    assert d['model'] is m
    return d['step'], d['leaves'], d['gout']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', nargs='?', default='BL2')
    ap.add_argument('batch', nargs='?', type=int, default=4)
    ap.add_argument('levels', nargs='?', type=int, default=4)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--profile-steps', type=int, default=0)
    ap.add_argument('--sites', action='store_true')
    ap.add_argument('--no-feat-grad', action='store_true', help='levels 1.. of the pyramid carry no gradient (rounds 3-5 protocol)')
    ap.add_argument('--checksum', action='store_true', help='print a checksum of every gradient (run-to-run stability)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    pc, m, cam, depth, ctx, mlvl = build(a.config, a.batch, a.levels, dev, feat_grad=not a.no_feat_grad)
    step, leaves, gout = make_step(m, cam, depth, ctx, mlvl, dev, pc, a.batch)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if a.profile_steps:
        for _ in range(a.profile_steps):
            step()
        torch.cuda.synchronize()
        print(json.dumps({'profiled_steps': a.profile_steps}))
        return
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = sorted(x.elapsed_time(y) for x, y in ev)
    # forward only, train mode (autograd graph recorded)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        o = m(cam, ctx, depth, mlvl_feats=mlvl)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / a.steps
    del o
    rec = {'config': a.config, 'B': a.batch, 'levels': a.levels, 'feat_grad': not a.no_feat_grad, 'ms_forward_backward_gradient_handed_over': 1e3 * el / a.steps,
           'gpu_ms_p10_p50_p90': [ms[int(0.1 * len(ms))], ms[len(ms) // 2], ms[min(len(ms) - 1, int(0.9 * len(ms)))]],
           'ms_forward_train_mode': 1e3 * tf, 'samples_per_s': a.batch * a.steps / el,
           'peak_mem_GB': torch.cuda.max_memory_allocated() / 2**30}
    if a.checksum:
        names = [n for n, _ in m.named_parameters()] + ['depth', 'ctx'] + [f'mlvl{i}' for i in range(1, len(mlvl or []))]
        step()
        a1 = [None if t.grad is None else t.grad.clone() for t in leaves]
        step()
        unstable = [n for n, x, t in zip(names, a1, leaves) if not ((x is None and t.grad is None) or torch.equal(x, t.grad))]
        rec['grad_bits_stable'] = not unstable
        rec['grad_bits_unstable'] = unstable
        rec['grad_abs_sums'] = {n: (None if t.grad is None else float(t.grad.double().abs().sum())) for n, t in zip(names, leaves)}
    if a.sites:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        rows = []
        for e in prof.key_averages(group_by_input_shape=True):
            if e.self_device_time_total > 20:
                rows.append({'op': e.key[:140], 'self_ms': e.self_device_time_total / 1e3, 'calls': e.count, 'shapes': str(e.input_shapes)[:160]})
        rows.sort(key=lambda r: -r['self_ms'])
        rec['op_sites'] = rows[:90]
        rec['launches'] = sum(e.count for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith('aten::')
                              and not e.key.endswith('Backward'))
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
