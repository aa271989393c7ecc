#!/usr/bin/env python3
"""Temporal history fusion at FB-OCC sizes: fb_bev_amd.TemporalHistoryFusion (HIP warp + folded GEMMs) vs the
reference's op sequence (fbocc.py:264-319: generate_grid, F.grid_sample, cats, Conv3d+BN+ReLU x2, clone) written in
plain torch on the same GPU.   python tools/time_history.py [Y X Z] [B] [f32|f16|bf16] [noref] [cbf16] [vm]
(f16 / bf16: the 16-bit history ring of BASELINE configs[4]; noref: skip the torch reference sequence; cbf16: the two
convolutions on the bf16 MFMA, history_compute=bfloat16; vm: ring_layout=voxel_major)"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd.history_fusion import TemporalHistoryFusion


def rigid(yaw, t):
    import math
    m = torch.eye(4)
    m[0, 0] = m[1, 1] = math.cos(yaw); m[0, 1] = -math.sin(yaw); m[1, 0] = math.sin(yaw)
    m[:3, 3] = torch.tensor(t)
    return m


class TorchReference:
    """the reference's sequence of torch ops, sharing the module's layers"""
    def __init__(self, m):
        self.m, self.hist, self.augs, self.sweep = m, None, None, None

    def generate_grid(self, hist_augs, fwd, ego, shape, device):
        n, _, z, h, w = shape
        m = self.m
        xs = torch.linspace(0, w - 1, w, device=device).view(1, w, 1).expand(h, w, z)
        ys = torch.linspace(0, h - 1, h, device=device).view(h, 1, 1).expand(h, w, z)
        zs = torch.linspace(0, z - 1, z, device=device).view(1, 1, z).expand(h, w, z)
        grid = torch.stack((xs, ys, zs, torch.ones_like(xs)), -1).view(1, h, w, z, 4).expand(n, h, w, z, 4).reshape(n, h, w, z, 4, 1)
        f2b = torch.zeros((4, 4), device=device)
        for i in range(3):
            f2b[i, i] = m.dx[i]; f2b[i, 3] = m.lower[i]
        f2b[3, 3] = 1
        f2b = f2b.view(1, 4, 4)
        flow = torch.inverse(f2b) @ hist_augs @ ego @ torch.inverse(fwd) @ f2b
        grid = flow.view(n, 1, 1, 1, 4, 4) @ grid
        nf = torch.tensor([w - 1.0, h - 1.0, z - 1.0], device=device)
        return grid[:, :, :, :, :3, 0] / nf.view(1, 1, 1, 1, 3) * 2.0 - 1.0

    def fuse(self, curr_bev, ego, bda, first=False):
        m = self.m; T, C = m.history_cat_num, m.single_bev_num_channels
        curr = curr_bev.permute(0, 1, 4, 2, 3)
        fwd = m.forward_augs(bda)
        if self.hist is None:
            self.hist = curr.repeat(1, T, 1, 1, 1); self.augs = fwd.clone(); self.sweep = curr.new_zeros(curr.shape[0], T)
        self.sweep = self.sweep + 1
        if first:
            self.sweep = torch.zeros_like(self.sweep)
        grid = self.generate_grid(self.augs, fwd, ego, curr.shape, curr.device)
        sampled = F.grid_sample(self.hist, grid.permute(0, 3, 1, 2, 4), align_corners=True, mode='bilinear')
        sweep = torch.cat([self.sweep.new_zeros(self.sweep.shape[0], 1), self.sweep], 1)
        feats_cat = torch.cat([curr, sampled], 1)
        n, _, z, h, w = feats_cat.shape
        f = feats_cat.reshape(n, T + 1, C, z, h, w)
        f = torch.cat([f, sweep[:, :, None, None, None, None].repeat(1, 1, 1, z, h, w) * m.history_cam_sweep_freq], 2)
        f = m.history_keyframe_time_conv(f.reshape(-1, C + 1, z, h, w)).reshape(n, T + 1, -1, z, h, w)
        out = m.history_keyframe_cat_conv(f.reshape(n, -1, z, h, w))
        self.hist = feats_cat[:, :-C].detach().clone(); self.sweep = sweep[:, :-1]; self.augs = fwd.clone()
        return out.permute(0, 1, 3, 4, 2).clone()


def main():
    Y, X, Z = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (100, 100, 8)
    B = int(sys.argv[4]) if len(sys.argv) >= 5 else 1
    C, T = 80, 16
    dt = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[5] if len(sys.argv) >= 6 else 'f32']
    noref = 'noref' in sys.argv
    comp = 'bf16x3' if 'cx3' in sys.argv else torch.bfloat16 if 'cbf16' in sys.argv else torch.float32
    lay = 'voxel_major' if 'vm' in sys.argv else 'planar'
    unfused = 'unfused' in sys.argv       # two-kernel path (warp, then convolutions) instead of fbbev_history_fused_vm
    esz = 4 if dt == torch.float32 else 2
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    dxv = 80.0 / X
    m = TemporalHistoryFusion([dxv, dxv, 6.4 / Z], [-40 + dxv / 2, -40 + dxv / 2, -1 + 3.2 / Z], C, T, history_dtype=dt,
                              history_compute=comp, ring_layout=lay).to(dev).eval()
    m.fused_warp_conv = not unfused
    if hasattr(m, 'fused_x3'):
        m.fused_x3 = os.environ.get('HIST_FUSED_X3', '1' if m.fused_x3 else '0') == '1'
    m.pipelined_step = os.environ.get('HIST_PIPE', '0') == '1'            # two-stream step (fbbev_history_step_x3_vm)
    m.pipelined_step_chunks = int(os.environ.get('HIST_CHUNKS', '0'))
    for seq in (m.history_keyframe_time_conv, m.history_keyframe_cat_conv):
        seq[1].running_var.uniform_(0.5, 1.5); seq[1].running_mean.uniform_(-0.2, 0.2)
    ref = TorchReference(m)
    ego_cpu = torch.stack([rigid(0.02 * (b + 1), [1.2, -0.1 * b, 0.02]) for b in range(B)])
    bda = torch.stack([rigid(0.1 * b, [0, 0, 0])[:3, :3] for b in range(B)]).to(dev)
    frames = [torch.randn(B, C, Y, X, Z, device=dev) for _ in range(3)]

    def metas(first):
        return [dict(sequence_group_idx=b, start_of_sequence=first, curr_to_prev_ego_rt=ego_cpu[b]) for b in range(B)]

    def timed(fn, n=8):
        ts = []
        for i in range(n + 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(i); b.record(); torch.cuda.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]

    with torch.no_grad():
        o1 = m.fuse_history(frames[0], metas(True), bda); o1 = m.fuse_history(frames[1], metas(False), bda)
        err = herr = t_ref = None
        t_hip = timed(lambda i: m.fuse_history(frames[i % 3], metas(False), bda))
        ego_dev = ego_cpu.to(dev)
        if not noref:
            r1 = ref.fuse(frames[0], ego_dev, bda, first=True); r1 = ref.fuse(frames[1], ego_dev, bda)
            m.reset(); o1 = m.fuse_history(frames[0], metas(True), bda); o1 = m.fuse_history(frames[1], metas(False), bda)
            err = (o1 - r1).abs().max().item()
            herr = (m.history_as_reference() - ref.hist).abs().max().item()
            t_ref = timed(lambda i: ref.fuse(frames[i % 3], ego_dev, bda))
        # the warp alone
        from fb_bev_amd import _capi
        hist = m.history_bev; flow = m.rt_flow(ego_dev, bda); dst = torch.empty_like(hist)
        if hist.dim() == 4:
            t_warp = timed(lambda i: _capi.history_warp_vm(hist, flow, dst, (Z, Y, X)))
        else:
            t_warp = timed(lambda i: _capi.history_warp(hist, flow, dst))
        rel = None
        if comp != torch.float32:          # same frames through the fp32 convolutions: what the reduced precision costs
            m.history_compute = torch.float32; m.reset()
            m.fuse_history(frames[0], metas(True), bda); o2 = m.fuse_history(frames[1], metas(False), bda)
            m.history_compute = comp; m.reset()
            m.fuse_history(frames[0], metas(True), bda); o3 = m.fuse_history(frames[1], metas(False), bda)
            rel = ((o3 - o2).abs().max() / o2.abs().max()).item()
    hist_bytes = B * T * C * Z * Y * X * esz
    print(json.dumps({'grid': [Y, X, Z], 'B': B, 'C': C, 'T': T, 'history_MB': round(hist_bytes / 1e6, 1),
                      'history_dtype': str(dt).split('.')[-1], 'conv_compute': str(comp).split('.')[-1], 'ring_layout': lay, 'fused_x3': bool(getattr(m, 'fused_x3', False)), 'pipelined_step': bool(m.pipelined_step), 'chunks': m.pipelined_step_chunks, 'warp_conv_one_kernel': bool(m.fused_warp_conv and comp == torch.bfloat16 and lay == 'voxel_major' and dt != torch.float32), 'fused_ms': round(t_hip, 4),
                      'torch_reference_sequence_ms': None if t_ref is None else round(t_ref, 4), 'speedup': None if t_ref is None else round(t_ref / t_hip, 2),
                      'warp_ms': round(t_warp, 4), 'warp_GBps_read_plus_write': round(2 * hist_bytes / t_warp / 1e6, 1),
                      'max_abs_diff_out': err, 'max_abs_diff_history': herr,
                      'bf16_convs_vs_fp32_convs_max_rel_to_peak': rel}))


if __name__ == '__main__':
    main()
