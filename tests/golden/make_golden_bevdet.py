#!/usr/bin/env python3
"""Fixture for the BEVDet-era view transformers: run the REAL `LSSViewTransformer` and
`LSSViewTransformer2` (mmdet3d/models/necks/view_transformer.py:16-329, 332-724) on CPU.  The file is loaded by path;
the bev_pool_v2 extension it calls is served by the C oracle (tests/golden/make_golden.py stand-ins), everything else --
geometry, ranking, the depth > 0.01 filter, the Z collapse -- is the reference's own code.

Run in the build container:  python tests/golden/make_golden_bevdet.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402


def main():
    MG.install_stubs()
    from fb_bev_amd import synthetic as S
    MG.load_ref('mmdet3d.ops.bev_pool_v2.bev_pool', 'mmdet3d/ops/bev_pool_v2/bev_pool.py')
    sys.modules['mmdet3d.models'].__path__ = []
    MG._mod('mmdet3d.models.necks').__path__ = []
    ref = MG.load_ref('mmdet3d.models.necks.view_transformer', 'mmdet3d/models/necks/view_transformer.py')
    torch.manual_seed(0)
    grid = {'x': [-8, 8, 1.0], 'y': [-8, 8, 1.0], 'z': [-1, 3, 2.0], 'depth': [1.0, 9.0, 1.0]}       # 16x16x2, D=8
    pc = S.PathConfig(name='bevdet', input_size=(64, 96), downsample=16, grid_config=grid, channels=8)
    B, N, Cin, C = 2, 6, 10, 8
    H, W = pc.feat_hw
    cam = S.camera_rig(pc, B, seed=0, bda_aug=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, Cin, H, W, generator=g) * 2.0          # peaky depth logits: a good share of bins is <= 0.01
    out = dict(x=x.numpy(), dims=np.array([B, N, Cin, C, H, W]))
    for i, t in enumerate(cam):
        out[f'cam{i}'] = t.numpy()
    for name, cls in (('v1', ref.LSSViewTransformer), ('v2', ref.LSSViewTransformer2)):
        vt = cls(grid_config=grid, input_size=(64, 96), downsample=16, in_channels=Cin, out_channels=C)
        with torch.no_grad():
            vt.depth_net.weight.normal_(0, 0.6)
            vt.depth_net.bias.normal_(0, 0.3)
            bev, depth = vt([x] + list(cam))
        out[f'{name}.w'], out[f'{name}.b'] = vt.depth_net.weight.detach().numpy(), vt.depth_net.bias.detach().numpy()
        out[f'{name}.bev'], out[f'{name}.depth'] = bev.numpy(), depth.numpy()
        print(name, tuple(bev.shape), 'kept share of depth bins:', float((depth > 0.01).float().mean()))
    path = os.path.join(MG.OUT, 'bevdet_view_transformer_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')
    make_trt_pool_fixture(sys.modules['mmdet3d.ops.bev_pool_v2.bev_pool'])


def make_trt_pool_fixture(ref_pool):
    """The REAL TRTBEVPoolv2.forward (ops/bev_pool_v2/bev_pool.py:118-141; imported at module scope by fbocc.py:12 and
    detectors/bevdet.py:6) on a Z = 1 grid: depth (n, d, h, w), feat, the index tensors of a single sample ->
    (1, out_height, out_width, C).  The extension under it is the C oracle (install_stubs)."""
    from fb_bev_amd import synthetic as S
    from oracle import oracle as O
    grid = {'x': [-8, 8, 1.0], 'y': [-6, 6, 1.0], 'z': [-10, 10, 20.0], 'depth': [1.0, 9.0, 1.0]}       # X=16, Y=12, Z=1
    pc = S.PathConfig(name='trt', input_size=(64, 96), downsample=16, grid_config=grid, channels=8)
    n, C = 6, 8
    H, W = pc.feat_hw
    D = pc.D
    cam = S.camera_rig(pc, 1, seed=2, bda_aug=True)
    ovt = O.ViewTransformerOracle(grid, pc.input_size, pc.downsample)
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(ovt.get_lidar_coor(*cam).contiguous())
    g = torch.Generator().manual_seed(3)
    depth = (torch.randn(n, D, H, W, generator=g) * 2).softmax(1).contiguous()
    # bev_pool.py:132 `feat.view(1, n, feat.shape[3], h, w)`: dim 3 of the argument is taken as the channel count and its MEMORY is
    # read in (n, C, h, w) order -- reproduced literally: a contiguous tensor shaped (n, h, w, C)
    feat_arg = torch.randn(n, H, W, C, generator=g).contiguous()
    out = ref_pool.TRTBEVPoolv2.forward(None, depth, feat_arg, rd, rf, rb, st, ln, 12, 16)
    path = os.path.join(MG.OUT, 'trt_bev_pool_v2_small.npz')
    np.savez_compressed(path, depth=depth.numpy(), feat=feat_arg.numpy(), ranks_depth=rd.numpy(), ranks_feat=rf.numpy(),
                        ranks_bev=rb.numpy(), interval_starts=st.numpy(), interval_lengths=ln.numpy(), out=out.numpy(),
                        out_hw=np.array([12, 16]))
    print('TRTBEVPoolv2', tuple(out.shape), 'wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
