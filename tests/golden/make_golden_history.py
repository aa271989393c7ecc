#!/usr/bin/env python3
"""Fixtures for the temporal history fusion: run the REAL FBOCC.fuse_history /
FBOCC.generate_grid (mmdet3d/models/fbbev/detectors/fbocc.py:169-319) on CPU over a short sequence.

The detector class cannot be constructed here (no mmdet/mmcv/spconv), so the module is loaded by path
with inert stand-ins for its third-party imports and the two methods run on an instance made with
object.__new__ whose attributes are exactly those FBOCC.__init__ sets for them (fbocc.py:101-131);
SyncBatchNorm -> BatchNorm3d in eval mode (same arithmetic with running statistics).
F.grid_sample is wrapped to record the grid and the sampled volume of every call.

Run in the build container:  python tests/golden/make_golden_history.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402


def install_history_stubs():
    MG.install_stubs()
    _mod = MG._mod
    sys.modules['mmcv.runner'].get_dist_info = lambda: (0, 1)
    sys.modules['mmdet.models'].DETECTORS = MG._Registry()
    _mod('mmdet.core', reduce_mean=lambda x: x)
    b = sys.modules['mmdet3d.models.builder']
    b.build_head = b.build_neck = b.build_backbone = lambda *a, **k: None
    sys.modules['mmdet3d.models'].builder = b

    class CenterPoint(nn.Module):
        pass
    _mod('mmdet3d.models.detectors', CenterPoint=CenterPoint)
    _mod('mmdet3d.models.fbbev.utils', run_time=lambda *a, **k: None)
    _mod('spconv'); _mod('spconv.pytorch')
    _mod('torchvision'); _mod('torchvision.utils', make_grid=None)
    _mod('mmdet3d.datasets'); _mod('mmdet3d.datasets.utils', nuscenes_get_rt_matrix=None)
    _mod('mmdet3d.core'); _mod('mmdet3d.core.bbox', box_np_ops=None)
    MG.load_ref('mmdet3d.ops.bev_pool_v2.bev_pool', 'mmdet3d/ops/bev_pool_v2/bev_pool.py')


def rz(deg):
    a = np.deg2rad(deg)
    return torch.tensor([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]], dtype=torch.float32)


def rigid(yaw_deg, t):
    m = torch.eye(4)
    m[:3, :3] = rz(yaw_deg)
    m[:3, 3] = torch.tensor(t, dtype=torch.float32)
    return m


def main(C=4, grid=(4, 10, 12), name='history_fusion_seq4.npz', full=True, seed=0):
    """full=False: the smaller record of the C = 16 sequence (what the voxel-major / MFMA routes need: inputs, fused output
    per frame, history state after the last frame) -- grid and sampled volume are pinned by the C = 4 fixture."""
    install_history_stubs()
    ref = MG.load_ref('refdet.fbocc', 'mmdet3d/models/fbbev/detectors/fbocc.py')
    FBOCC = ref.FBOCC
    torch.manual_seed(seed)
    B, T = 2, 3
    Z, Y, X = grid
    det = object.__new__(FBOCC)
    nn.Module.__init__(det)
    # fbocc.py:101-131 (3-D grid => Conv3d); forward_projection only contributes dx / bx (view_transformer.py gen_dx_bx)
    dx = torch.tensor([0.8, 0.8, 0.8]); lo = torch.tensor([-4.8, -4.0, -1.0])
    det.forward_projection = types.SimpleNamespace(dx=dx, bx=lo + dx / 2.0, nx=torch.tensor([X, Y, Z]))
    det.single_bev_num_channels = C
    det.do_history = True
    det.interpolation_mode = 'bilinear'
    det.history_cat_num = T
    det.history_cam_sweep_freq = 0.5
    det.history_keyframe_time_conv = nn.Sequential(nn.Conv3d(C + 1, C, 1), nn.BatchNorm3d(C), nn.ReLU(inplace=True))
    det.history_keyframe_cat_conv = nn.Sequential(nn.Conv3d(C * (T + 1), C, 1), nn.BatchNorm3d(C), nn.ReLU(inplace=True))
    for seq in (det.history_keyframe_time_conv, det.history_keyframe_cat_conv):
        bn = seq[1]
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.6, 1.4)
        bn.weight.data.uniform_(0.7, 1.3); bn.bias.data.uniform_(-0.1, 0.1)
    det.eval()
    det.history_sweep_time = None
    det.history_bev = None
    det.history_seq_ids = None
    det.history_forward_augs = None

    rec = {}
    real_gs = torch.nn.functional.grid_sample

    def spy(inp, grid, **kw):
        out = real_gs(inp, grid, **kw)
        rec['grid'], rec['sampled'], rec['kw'] = grid.clone(), out.clone(), dict(kw)
        return out
    ref.F.grid_sample = spy

    frames = [   # (start_of_sequence per sample, seq ids, ego motion curr->prev per sample, bda per sample)
        dict(start=[True, True], seq=[3, 7], ego=[rigid(0, [0, 0, 0]), rigid(0, [0, 0, 0])], bda=[rz(0), rz(10)]),
        dict(start=[False, False], seq=[3, 7], ego=[rigid(4.0, [0.9, -0.3, 0.05]), rigid(-7.0, [1.7, 0.4, 0.0])],
             bda=[rz(-12) @ torch.diag(torch.tensor([1., -1., 1.])), rz(5)]),
        dict(start=[False, True], seq=[3, 9], ego=[rigid(2.0, [1.1, 0.2, -0.04]), rigid(0, [0, 0, 0])],
             bda=[rz(20), rz(-15) @ torch.diag(torch.tensor([-1., 1., 1.]))]),
        dict(start=[False, False], seq=[3, 9], ego=[rigid(-3.0, [0.7, 0.1, 0.02]), rigid(6.0, [1.3, -0.6, 0.03])],
             bda=[rz(0), rz(0)]),
    ]
    out = {'dims': np.array([B, C, T, Z, Y, X]), 'dx': dx.numpy(), 'bx': det.forward_projection.bx.numpy()}
    for k, v in det.state_dict().items():
        out['w.' + k] = v.numpy()
    with torch.no_grad():
        for i, f in enumerate(frames):
            curr = torch.randn(B, C, Y, X, Z)                    # the (B,C,Y,X,Z) view the view transformer returns
            metas = [dict(sequence_group_idx=f['seq'][b], start_of_sequence=f['start'][b],
                          curr_to_prev_ego_rt=f['ego'][b]) for b in range(B)]
            bda = torch.stack(f['bda'])
            res = det.fuse_history(curr.clone(), metas, bda)
            out[f'f{i}.curr'] = curr.numpy()
            out[f'f{i}.bda'] = bda.numpy()
            out[f'f{i}.ego'] = torch.stack(f['ego']).numpy()
            out[f'f{i}.seq'] = np.array(f['seq']); out[f'f{i}.start'] = np.array(f['start'])
            out[f'f{i}.out'] = res.numpy()
            if full:
                out[f'f{i}.grid'] = rec['grid'].numpy()              # (B,Z,Y,X,3) normalised sampling grid
                out[f'f{i}.sampled'] = rec['sampled'].numpy()        # (B,T*C,Z,Y,X)
            if full or i == len(frames) - 1:
                out[f'f{i}.history_after'] = det.history_bev.clone().numpy()   # clone: the state is mutated in place next frame
            out[f'f{i}.sweep_time_after'] = det.history_sweep_time.clone().numpy()
            assert rec['kw'] == {'align_corners': True, 'mode': 'bilinear'}
    path = os.path.join(MG.OUT, name)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
    # C = 16: the channel count the register-resident MFMA convolutions and the voxel-major ring take
    main(C=16, grid=(3, 8, 9), name='history_fusion_seq4_c16.npz', full=False, seed=1)
