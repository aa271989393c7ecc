"""GPU run of the BEVDet-era view transformers against the real-reference fixture.  The CPU suite checks the same chain
on the emulated kernels (tests/test_bevdet_view_transformer.py); first passed on an MI355X in round 2
(gpurun_out/s1_gated_tests.log) and part of the default GPU suite since."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu]
G = os.path.join(os.path.dirname(__file__), 'golden', 'bevdet_view_transformer_small.npz')
GRID = {'x': [-8, 8, 1.0], 'y': [-8, 8, 1.0], 'z': [-1, 3, 2.0], 'depth': [1.0, 9.0, 1.0]}


@pytest.mark.parametrize('name', ['v1', 'v2'])
def test_module_equals_reference_fixture_and_backpropagates(name):
    from fb_bev_amd.bevdet_view_transformer import LSSViewTransformer, LSSViewTransformer2
    dev = torch.device('cuda:0')
    gold = np.load(G)
    B, N, Cin, C, H, W = gold['dims'].tolist()
    m = (LSSViewTransformer if name == 'v1' else LSSViewTransformer2)(grid_config=GRID, input_size=(64, 96), downsample=16,
                                                                       in_channels=Cin, out_channels=C).to(dev)
    with torch.no_grad():
        m.depth_net.weight.copy_(torch.from_numpy(gold[f'{name}.w']))
        m.depth_net.bias.copy_(torch.from_numpy(gold[f'{name}.b']))
    inp = [torch.from_numpy(gold['x']).to(dev)] + [torch.from_numpy(gold[f'cam{i}']).to(dev) for i in range(6)]
    with torch.no_grad():
        bev, depth = m(inp)
    assert torch.allclose(bev.cpu(), torch.from_numpy(gold[f'{name}.bev']), atol=1e-5, rtol=1e-5)
    inp[0].requires_grad_()
    bev, _ = m(inp)
    bev.square().sum().backward()
    assert inp[0].grad is not None and torch.isfinite(inp[0].grad).all() and m.depth_net.weight.grad.abs().sum() > 0


def test_trt_bev_pool_v2_forward_equals_the_real_class_fixture():
    """fb_bev_amd.bev_pool.TRTBEVPoolv2.forward on the HIP op == the REAL TRTBEVPoolv2.forward (ops/bev_pool_v2/bev_pool.py:118-141;
    fixture of tests/golden/make_golden_bevdet.py, the extension under the real class served by the C oracle): bit for bit."""
    from fb_bev_amd.bev_pool import TRTBEVPoolv2
    dev = torch.device('cuda:0')
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trt_bev_pool_v2_small.npz'))
    t = {k: torch.from_numpy(z[k]).to(dev) for k in z.files if k != 'out_hw'}
    oh, ow = z['out_hw'].tolist()
    out = TRTBEVPoolv2.apply(t['depth'], t['feat'], t['ranks_depth'], t['ranks_feat'], t['ranks_bev'], t['interval_starts'],
                             t['interval_lengths'], oh, ow)
    assert out.shape == (1, oh, ow, t['feat'].shape[3])
    assert torch.equal(out.cpu(), torch.from_numpy(z['out']))
