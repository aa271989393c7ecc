"""CM_DepthNet against the fixture produced by the REAL reference module
(tests/golden/make_golden_depthnet.py; depth_net.py:258-446).  The net is vendor-library PyTorch (MIOpen / hipBLASLt
on the GPU), so the same module is checked on CPU here and on the GPU in test_gpu_depth_net."""
import os

import numpy as np
import pytest
import torch

from fb_bev_amd.depth_net import CM_DepthNet

G = os.path.join(os.path.dirname(__file__), 'golden', 'depth_net_small.npz')


def _load(device='cpu', **kw):
    z = np.load(G)
    B, N, Cin, H, W = (int(v) for v in z['dims'])
    net = CM_DepthNet(in_channels=Cin, context_channels=8, depth_channels=12, mid_channels=32, use_dcn=False, downsample=4,
                      grid_config=dict(depth=[1.0, 13.0, 1.0]), loss_depth_weight=1.0, **kw)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    missing, unexpected = net.load_state_dict(sd, strict=True), None      # the reference's own key names
    return z, net.to(device).eval()


def _t(z, k, device='cpu'):
    return torch.from_numpy(z[k]).to(device)


@pytest.mark.parametrize('channels_last', [True, False])
def test_forward_matches_reference_fixture(channels_last):
    z, net = _load(channels_last=channels_last)
    with torch.no_grad():
        mlp = net.get_mlp_input(*(_t(z, k) for k in ('rot', 'tran', 'intrin', 'post_rot', 'post_tran', 'bda')))
        assert torch.equal(mlp, _t(z, 'mlp_input'))
        context, depth = net(_t(z, 'x'), mlp)
    assert context.shape == z['context'].shape and depth.shape == z['depth'].shape
    assert torch.allclose(context, _t(z, 'context'), atol=1e-5, rtol=1e-5)
    assert torch.allclose(depth, _t(z, 'depth'), atol=1e-6, rtol=1e-5)
    assert torch.allclose(depth.sum(2), torch.ones_like(depth.sum(2)), atol=1e-5)       # a distribution over the D bins


def test_depth_supervision_matches_reference_fixture():
    z, net = _load()
    labels = net.get_downsampled_gt_depth(_t(z, 'gt'))
    assert torch.equal(labels, _t(z, 'labels'))
    loss = net.get_depth_loss(_t(z, 'gt'), _t(z, 'depth'))['loss_depth']
    assert abs(float(loss) - float(z['loss'])) < 1e-4 * max(1.0, abs(float(z['loss'])))


def test_shipped_config_block_builds_and_use_dcn_is_rejected():
    net = CM_DepthNet(in_channels=256, context_channels=80, downsample=16,
                      grid_config={'depth': [2.0, 42.0, 0.5]}, depth_channels=80, with_cp=False, loss_depth_weight=1.,
                      use_dcn=False)              # the depth_net block of fbocc-r50-cbgs_depth_16f_16x4_20e.py:138-148
    assert net.depth_conv[-1].out_channels == 80 and net.context_conv.out_channels == 80
    assert sum(p.numel() for p in net.parameters()) > 20e6
    with pytest.raises(NotImplementedError):
        CM_DepthNet(use_dcn=True)


@pytest.mark.gpu
def test_gpu_depth_net_matches_fixture_and_feeds_the_lift_splat():
    dev = torch.device('cuda:0')
    z, net = _load(dev)
    with torch.no_grad():
        mlp = net.get_mlp_input(*(_t(z, k, dev) for k in ('rot', 'tran', 'intrin', 'post_rot', 'post_tran', 'bda')))
        context, depth = net(_t(z, 'x', dev), mlp)
    assert (context.cpu() - _t(z, 'context')).abs().max().item() < 1e-4
    assert (depth.cpu() - _t(z, 'depth')).abs().max().item() < 1e-5
    # bf16 compute option: same distribution within bf16 accuracy, outputs still fp32
    z, net16 = _load(dev, compute_dtype=torch.bfloat16)
    with torch.no_grad():
        c16, d16 = net16(_t(z, 'x', dev), mlp)
    assert c16.dtype == torch.float32 and d16.dtype == torch.float32
    assert (d16 - depth).abs().max().item() < 0.05 and (c16 - context).abs().max().item() < 0.5
