"""BEVDet-era view transformers against a fixture of the REAL reference classes
(tests/golden/make_golden_bevdet.py: LSSViewTransformer, LSSViewTransformer2 of mmdet3d/models/necks/view_transformer.py).
The module's GPU stages are served by the same kernels on the CPU device emulator, so the chain
depth_net -> geometry -> (depth-thresholded) ranking -> fused dense pooling -> Z collapse is checked end to end."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))
import emu_capi as E  # noqa: E402

G = os.path.join(os.path.dirname(__file__), 'golden', 'bevdet_view_transformer_small.npz')
GRID = {'x': [-8, 8, 1.0], 'y': [-8, 8, 1.0], 'z': [-1, 3, 2.0], 'depth': [1.0, 9.0, 1.0]}


def _on_emulator(m):
    """Replace the four GPU entry points of the module by the emulated kernels (same C ABI, CPU tensors)."""
    dev = torch.device('cpu')

    def pack(t):
        rb, rd, rf, st, ln, ir, counts = t
        return types.SimpleNamespace(rb=rb, rd=rd, rf=rf, st=st, ln=ln, ir=ir, counts=counts)
    m.get_lidar_coor = lambda *cam: E.lidar_coor(*m._axes(dev), [c.contiguous().float() for c in cam])
    m.build_index = lambda coor, depth=None, depth_threshold=0.01: pack(E.rank_build(
        coor.contiguous(), *m._grid3(), depth=None if depth is None else depth.contiguous().float(), depth_threshold=depth_threshold))
    m.build_index_from_cams = lambda *cam, cached=False: pack(E.lift_rank_build(*m._axes(dev), [c.contiguous().float() for c in cam], *m._grid3())[:7])

    def lift_splat(idx, depth, feat):
        Z, Y, X = m.grid_zyx
        B, C = depth.shape[0], feat.shape[2]
        code, out = E.pool_dense(depth.contiguous().float(), feat.permute(0, 1, 3, 4, 2).contiguous().float(), idx.rd, idx.rf, idx.ir,
                                 idx.st, idx.ln, idx.counts, idx.st.numel(), B, C, Z, Y, X, 64, 0)
        assert code == 0 and not torch.isnan(out).any()
        return out.permute(0, 1, 3, 4, 2)
    m.lift_splat = lift_splat
    return m


@pytest.mark.parametrize('name', ['v1', 'v2'])
def test_module_equals_reference_fixture(name):
    from fb_bev_amd.bevdet_view_transformer import LSSViewTransformer, LSSViewTransformer2
    gold = np.load(G)
    B, N, Cin, C, H, W = gold['dims'].tolist()
    cls = LSSViewTransformer if name == 'v1' else LSSViewTransformer2
    m = cls(grid_config=GRID, input_size=(64, 96), downsample=16, in_channels=Cin, out_channels=C)
    assert list(m.state_dict()) == ['dx', 'bx', 'nx', 'depth_net.weight', 'depth_net.bias']
    with torch.no_grad():
        m.depth_net.weight.copy_(torch.from_numpy(gold[f'{name}.w']))
        m.depth_net.bias.copy_(torch.from_numpy(gold[f'{name}.b']))
    _on_emulator(m)
    cam = [torch.from_numpy(gold[f'cam{i}']) for i in range(6)]
    with torch.no_grad():
        bev, depth = m([torch.from_numpy(gold['x'])] + cam)
        bev2, depth2, digit = m([torch.from_numpy(gold['x'])] + cam, return_depth_digit=True)
    assert torch.allclose(depth, torch.from_numpy(gold[f'{name}.depth']), atol=1e-6)
    exp = torch.from_numpy(gold[f'{name}.bev'])
    assert bev.shape == exp.shape == (B, 2 * C, 16, 16)
    # the reference pools in the order of an unstable argsort; fp32 sums of a voxel's points may differ in the last bits
    assert torch.allclose(bev, exp, atol=1e-5, rtol=1e-5), (bev - exp).abs().max()
    assert torch.equal(bev, bev2) and digit.shape == (B * N, 8, H, W)
    if name == 'v2':        # the threshold matters on this input: the two fixtures differ
        assert not np.allclose(gold['v1.bev'], gold['v2.bev'], atol=1e-4)


def test_onnx_symbolic_emits_the_reference_node():
    """TRTBEVPoolv2.symbolic (ops/bev_pool_v2/bev_pool.py:94-116): node name, input order and attribute names."""
    from fb_bev_amd.bev_pool import TRTBEVPoolv2

    class FakeGraph:
        def op(self, name, *inputs, **attrs):
            return name, inputs, attrs
    name, inputs, attrs = TRTBEVPoolv2.symbolic(FakeGraph(), 'depth', 'feat', 'rd', 'rf', 'rb', 'starts', 'lengths', 200, 100)
    assert name == 'mmdeploy::bev_pool_v2'
    assert inputs == ('depth', 'feat', 'rd', 'rf', 'rb', 'starts', 'lengths')
    assert attrs == {'out_height_i': 200, 'out_width_i': 100}


def test_trt_bev_pool_v2_forward_equals_the_real_class_fixture(monkeypatch):
    """TRTBEVPoolv2.forward (ops/bev_pool_v2/bev_pool.py:118-141) against a fixture of the REAL class
    (tests/golden/make_golden_bevdet.py): the host logic of fb_bev_amd.bev_pool.TRTBEVPoolv2 with the extension call served by the
    emulated kernel (the GPU twin of this test is tests/test_gpu_bevdet.py)."""
    from fb_bev_amd import bev_pool, bev_pool_v2_ext
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trt_bev_pool_v2_small.npz'))

    def fwd(depth, feat, out, rd, rf, rb, lengths, starts):
        E.pool_fwd(depth, feat, out, rd, rf, rb, starts, lengths)
    monkeypatch.setattr(bev_pool_v2_ext, 'bev_pool_v2_forward', fwd)
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    oh, ow = z['out_hw'].tolist()
    out = bev_pool.TRTBEVPoolv2.forward(None, t['depth'], t['feat'], t['ranks_depth'], t['ranks_feat'], t['ranks_bev'],
                                        t['interval_starts'], t['interval_lengths'], oh, ow)
    assert out.shape == t['out'].shape == (1, oh, ow, t['feat'].shape[3])
    assert torch.equal(out, t['out'])                      # the same in-order fmaf chain as the oracle that served the real class
    assert 'TRTBEVPoolv2' in bev_pool.__all__
