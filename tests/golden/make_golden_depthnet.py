#!/usr/bin/env python3
"""Fixture for the camera-aware depth net: run the REAL CM_DepthNet
(mmdet3d/models/fbbev/modules/depth_net.py:258-446) on CPU at small sizes.

The module file is loaded by path with inert stand-ins for mmcv / mmdet / torchvision / cv2.  One stand-in carries
arithmetic: mmdet's ResNet `BasicBlock` (external, not in the tree) is served by fb_bev_amd.depth_net.BasicBlock -- so the
fixture pins everything EXCEPT that block (conv3x3-BN-ReLU-conv3x3-BN + identity, ReLU, restated from mmdet).

Run in the build container:  python tests/golden/make_golden_depthnet.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402


def main():
    MG.install_stubs()
    from fb_bev_amd.depth_net import BasicBlock
    sys.modules['mmdet.models.backbones.resnet'].BasicBlock = BasicBlock
    sys.modules['mmdet.models'].HEADS = MG._Registry()
    sys.modules['mmdet3d.models'].builder = sys.modules['mmdet3d.models.builder']
    MG._mod('torchvision'); MG._mod('torchvision.utils', make_grid=None)
    ref = MG.load_ref('refmod.depth_net', 'mmdet3d/models/fbbev/modules/depth_net.py')
    torch.manual_seed(0)
    B, N, Cin, H, W = 2, 3, 16, 4, 6
    grid = dict(depth=[1.0, 13.0, 1.0])
    net = ref.CM_DepthNet(in_channels=Cin, context_channels=8, depth_channels=12, mid_channels=32, use_dcn=False, downsample=4,
                          grid_config=grid, loss_depth_weight=1.0)
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.7, 1.3); m.bias.data.uniform_(-0.2, 0.2)
    net.eval()
    x = torch.randn(B, N, Cin, H, W)
    g = torch.Generator().manual_seed(1)
    rot = torch.randn(B, N, 3, 3, generator=g); tran = torch.randn(B, N, 3, generator=g)
    intrin = torch.randn(B, N, 3, 3, generator=g) * 100; post_rot = torch.randn(B, N, 3, 3, generator=g)
    post_tran = torch.randn(B, N, 3, generator=g); bda = torch.randn(B, 3, 3, generator=g)
    with torch.no_grad():
        mlp_in = net.get_mlp_input(rot, tran, intrin, post_rot, post_tran, bda)
        context, depth = net(x, mlp_in)
        gt = torch.rand(B, N, H * 4, W * 4, generator=g) * 14.0
        gt[torch.rand(gt.shape, generator=g) < 0.5] = 0.0
        labels = net.get_downsampled_gt_depth(gt)
        loss = net.get_depth_loss(gt, depth)['loss_depth']
    out = dict(x=x.numpy(), rot=rot.numpy(), tran=tran.numpy(), intrin=intrin.numpy(), post_rot=post_rot.numpy(),
               post_tran=post_tran.numpy(), bda=bda.numpy(), mlp_input=mlp_in.numpy(), context=context.numpy(),
               depth=depth.numpy(), gt=gt.numpy(), labels=labels.numpy(), loss=np.array(float(loss)),
               dims=np.array([B, N, Cin, H, W]))
    for k, v in net.state_dict().items():
        out['w.' + k] = v.numpy()
    path = os.path.join(MG.OUT, 'depth_net_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(net.state_dict()), 'state entries')


if __name__ == '__main__':
    main()
