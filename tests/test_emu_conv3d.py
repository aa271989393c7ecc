"""fbbev_conv3d_ndhwc (fp32-MFMA implicit-GEMM 3-D convolution, csrc/conv3d_kernels.h) on the CPU device emulator
against torch's fp32 convolutions, and the eval-mode mapping of CustomResNet3D / FPN3D / OccHead onto it
(fb_bev_amd/mfma_conv3d.py) against the modules' own forward.  Tolerance 1e-4 abs on O(1) activations: both sides are
fp32, only the summation order over (tap, cin) differs."""
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))
import emu_capi as E  # noqa: E402
from fb_bev_amd import mfma_conv3d as M  # noqa: E402


def emu_backend(x, wf, bias, out, Cout, planar=False, tiled=False, ksize=3, stride=1, pad=1, relu=False, residual=None,
                transposed=False):
    """The HIP entry points on the emulator, chosen like mfma_conv3d._launch: by the weight layout, `planar` and `tiled`."""
    res = None if residual is None else residual.contiguous()
    if tiled:
        assert wf.dtype == torch.bfloat16 and (ksize, stride, pad, transposed, planar) == (3, 1, 1, False, False)
        code, y = E.conv3d_k3s1_tiled_bf16(x.contiguous(), wf, bias, Cout, relu=relu, residual=res)
    elif wf.dtype == torch.bfloat16:
        x5 = x.unsqueeze(1) if planar else x
        code, y = E.conv3d_ndhwc_bf16(x5.contiguous(), wf, bias, Cout, ksize=ksize, stride=stride, pad=pad, relu=relu,
                                      residual=None if res is None else (res.unsqueeze(1) if planar else res), transposed=transposed,
                                      planar=planar)
        y = y.squeeze(1) if planar else y
    elif planar:
        code, y = E.conv2d_nhwc(x.contiguous(), wf, bias, Cout, ksize=ksize, stride=stride, pad=pad, relu=relu, residual=res)
    else:
        code, y = E.conv3d_ndhwc(x.contiguous(), wf, bias, Cout, ksize=ksize, stride=stride, pad=pad, relu=relu, residual=res,
                                 transposed=transposed)
    assert code == 0
    assert tuple(y.shape) == tuple(out.shape) and not torch.isnan(y).any()          # every element written
    return y


@pytest.mark.parametrize('B,dims,Cin,Cout,k,s,p,relu,res', [
    (1, (5, 6, 3), 16, 16, 3, 1, 1, False, False),         # MT=1
    (2, (5, 6, 3), 32, 64, 3, 2, 1, True, False),          # MT=4, stride 2, two samples, odd extents
    (1, (4, 9, 8), 16, 32, 3, 1, 1, True, True),           # MT=2, 288 voxels: two workgroups, partial last wave; residual
    (1, (3, 5, 2), 48, 19, 1, 1, 0, False, False),         # 1x1x1 to 19 classes: scalar store path, padded cout tile
    (1, (6, 6, 4), 16, 80, 1, 2, 0, False, False),         # strided 1x1x1 (downsample branch), 5 cout tiles
    (1, (3, 4, 2), 32, 4, 1, 1, 0, False, True),           # 4 soft-weight channels
])
def test_conv3d_kernel_vs_torch(B, dims, Cin, Cout, k, s, p, relu, res):
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = torch.randn(B, Cin, *dims, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g)
    exp = F.conv3d(x, w, b, stride=s, padding=p)
    r = torch.randn(exp.shape, generator=g) if res else None
    if res:
        exp = exp + r
    if relu:
        exp = exp.relu()
    wf = M.weight_fragments(w)
    bias = F.pad(b, (0, (Cout + 15) // 16 * 16 - Cout))
    code, y = E.conv3d_ndhwc(M.to_ndhwc(x), wf, bias, Cout, ksize=k, stride=s, pad=p, relu=relu,
                             residual=None if r is None else M.to_ndhwc(r))
    assert code == 0 and not torch.isnan(y).any()
    assert torch.allclose(M.to_ncdhw(y), exp, atol=1e-4, rtol=1e-4), (M.to_ncdhw(y) - exp).abs().max()


def test_transposed_conv_k2s2_vs_torch():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 3, 4, 2, generator=g)
    w = torch.randn(32, 24, 2, 2, 2, generator=g) / 32 ** 0.5
    exp = F.conv_transpose3d(x, w, None, stride=2).relu()
    wf = M.weight_fragments(w, transposed=True)
    code, y = E.conv3d_ndhwc(M.to_ndhwc(x), wf, torch.zeros(32), 24, relu=True, transposed=True)
    assert code == 0 and not torch.isnan(y).any()
    assert torch.allclose(M.to_ncdhw(y), exp, atol=1e-4, rtol=1e-4)


def test_conv3d_argument_checks():
    x = torch.zeros(1, 2, 2, 2, 16)
    wf = torch.zeros(27 * 256)
    bias = torch.zeros(16)
    assert E.conv3d_ndhwc(torch.zeros(1, 2, 2, 2, 8), wf, bias, 16)[0] == -2                # Cin % 16
    assert E.conv3d_ndhwc(x, wf, bias, 16, ksize=5, pad=2)[0] == -2
    out = torch.zeros(1, 3, 2, 2, 16)                                                        # wrong Do
    code = E.lib().fbbev_conv3d_ndhwc(E.p(x), E.p(wf), E.p(bias), None, 1, 2, 2, 2, 16, 3, 2, 2, 16, 3, 1, 1, 0, 0, E.p(out), None)
    assert code == -1


@pytest.mark.parametrize('B,dims,Cin,Cout,k,s,p', [
    (1, (5, 6, 3), 16, 16, 3, 1, 1),
    (2, (6, 6, 4), 16, 32, 3, 2, 1),          # stride 2: taps selected by parity
    (1, (5, 7, 3), 32, 16, 3, 2, 1),          # odd extents: the last input plane gets no gradient from some taps
    (1, (4, 4, 2), 48, 16, 1, 1, 0),
    (1, (6, 4, 4), 16, 80, 1, 2, 0),          # strided 1x1x1: 7/8 of the dx voxels are exactly zero
    (2, (3, 4, 2), 16, 32, 2, 2, 0),          # kernel 2 stride 2 (the data gradient of the head's deconvolution, see below)
])
def test_conv3d_dgrad_and_wgrad_vs_torch_autograd(B, dims, Cin, Cout, k, s, p):
    g = torch.Generator().manual_seed(Cin + 7 * Cout + k)
    dims = tuple(d if (d + 2 * p - k) // s + 1 > 0 else k for d in dims)
    if k == 2:
        dims = tuple(2 * ((d + 1) // 2) for d in dims)                  # kernel 2 stride 2 tiles the input exactly
    x = torch.randn(B, Cin, *dims, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5).requires_grad_()
    y = F.conv3d(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    code, dx = E.conv3d_dgrad_ndhwc(M.to_ndhwc(dy), M.weight_fragments(w.detach().transpose(0, 1)), dims, Cin, ksize=k, stride=s, pad=p)
    assert code == 0 and not torch.isnan(dx).any()
    assert torch.allclose(M.to_ncdhw(dx), x.grad, atol=1e-4, rtol=1e-4), (M.to_ncdhw(dx) - x.grad).abs().max()
    code, dw = E.conv3d_wgrad_ndhwc(M.to_ndhwc(x.detach()), M.to_ndhwc(dy), ksize=k, stride=s, pad=p)
    assert code == 0
    got = dw.view(k, k, k, Cout, Cin).permute(3, 4, 0, 1, 2)
    assert torch.allclose(got, w.grad, atol=2e-4, rtol=1e-4), (got - w.grad).abs().max()


def test_wgrad_many_chunks_and_channel_tails():
    """> 256 voxels per chunk boundary, Cout / Cin not multiples of 64 (80 and 20), two samples."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 20, 12, 10, 6, generator=g, requires_grad=False)
    w = torch.randn(80, 20, 3, 3, 3, generator=g, requires_grad=True)
    y = F.conv3d(x, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    code, dw = E.conv3d_wgrad_ndhwc(M.to_ndhwc(x), M.to_ndhwc(dy))
    assert code == 0
    got = dw.view(3, 3, 3, 80, 20).permute(3, 4, 0, 1, 2)
    assert torch.allclose(got, w.grad, atol=1e-3, rtol=1e-4), (got - w.grad).abs().max()
    assert E.conv3d_wgrad_ndhwc(torch.zeros(1, 2, 2, 2, 6), torch.zeros(1, 2, 2, 2, 8))[0] == -2      # Cin % 4


def emu_blend(level0, coarse, wsoft, out):
    code, y = E.blend_levels_ndhwc(level0.contiguous(), coarse, wsoft)
    assert code == 0 and not torch.isnan(y).any()
    return y


@pytest.mark.parametrize('dims,coarse_dims,C,K', [((8, 6, 4), [(4, 3, 2), (2, 2, 1)], 8, 3), ((10, 10, 4), [(5, 5, 2), (3, 3, 1), (1, 2, 1)], 16, 4),
                                               ((4, 4, 2), [], 4, 1)])
def test_blend_levels_kernel_vs_torch_interpolate(dims, coarse_dims, C, K):
    g = torch.Generator().manual_seed(C)
    B = 2
    level0 = torch.randn(B, *dims, C, generator=g)
    coarse = [torch.randn(B, *cd, C, generator=g) for cd in coarse_dims]
    w = torch.rand(B, *dims, K, generator=g).softmax(-1)
    exp = level0 * w[..., :1]
    for k, f in enumerate(coarse):
        up = F.interpolate(M.to_ncdhw(f), size=list(dims), mode='trilinear', align_corners=False).permute(0, 2, 3, 4, 1)
        exp = exp + up * w[..., k + 1:k + 2]
    code, got = E.blend_levels_ndhwc(level0, coarse, w)
    assert code == 0 and not torch.isnan(got).any()
    assert torch.allclose(got, exp, atol=2e-6, rtol=1e-5), (got - exp).abs().max()


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.6 - 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.6 + 0.7)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.4 - 0.2)
    return net.eval()


def test_voxel_encoder_and_head_runners_equal_the_modules():
    """Eval-mode CustomResNet3D -> FPN3D -> OccHead through the folded single-launch groups == the torch modules."""
    from fb_bev_amd.bev_encoder import CustomResNet3D, FPN3D
    from fb_bev_amd.occ_head import OccHead
    torch.manual_seed(0)
    chans = [16, 32, 48]
    bb = _randomise(CustomResNet3D(depth=10, block_strides=[1, 2, 2], n_input_channels=16, block_inplanes=chans,
                                   out_indices=(0, 1, 2), norm_cfg=dict(type='SyncBN')), 1)
    neck = _randomise(FPN3D(in_channels=chans, out_channels=64, norm_cfg=dict(type='SyncBN')), 2)
    head = _randomise(OccHead(in_channels=[64] * 3, out_channel=19, num_level=3, soft_weights=True, use_focal_loss=False,
                              norm_cfg=dict(type='SyncBN'), final_occ_size=[16, 16, 8], empty_idx=18), 3)
    x = torch.randn(1, 16, 8, 8, 4)
    with torch.no_grad():
        f_ref = bb(x)
        n_ref = neck(f_ref)
        o_ref = head(n_ref)['output_voxels'][0]
        f = M.ResNet3DRunner(bb)(M.to_ndhwc(x), backend=emu_backend)
        n = M.FPN3DRunner(neck)(f, backend=emu_backend)
        o = M.OccHeadRunner(head)(n, backend=emu_backend, blend_backend=emu_blend)
    for a, b in zip(f, f_ref):
        assert torch.allclose(M.to_ncdhw(a), b, atol=1e-4, rtol=1e-4)
    for a, b in zip(n, n_ref):
        assert torch.allclose(M.to_ncdhw(a), b, atol=1e-4, rtol=1e-4)
    assert o.shape == o_ref.shape == (1, 19, 16, 16, 8)
    assert torch.allclose(o, o_ref, atol=2e-4, rtol=1e-4), (o - o_ref).abs().max()


def test_groupnorm_stacks_are_rejected():
    from fb_bev_amd.bev_encoder import FPN3D
    with pytest.raises(NotImplementedError):
        M.FPN3DRunner(FPN3D(in_channels=[32], out_channels=32, norm_cfg=dict(type='GN', num_groups=4)))


def test_detector_opt_in_route_equals_module_route(monkeypatch):
    """FBOCC(execution=dict(mfma_conv3d=True)).predict_occupancy through the folded MFMA stacks (emulated) == the torch
    modules; the two GPU-only stages in front are replaced by CPU stand-ins as in tests/test_fbocc_model.py."""
    from fb_bev_amd import _capi, synthetic as S
    from fb_bev_amd.fbocc import FBOCC
    grid = {'x': [-8, 8, 2.0], 'y': [-8, 8, 2.0], 'z': [-1, 2.2, 0.8], 'depth': [2.0, 10.0, 1.0]}     # 8x8x4, D=8
    C = 16
    cfg = dict(
        fix_void=True, do_history=True, history_cat_num=2, single_bev_num_channels=C, readd=True,
        img_backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(2, 3), norm_eval=False, base_channels=8),
        img_neck=dict(type='CustomFPN', in_channels=[32, 64], out_channels=24, num_outs=1, start_level=0, out_ids=[0]),
        depth_net=dict(type='CM_DepthNet', in_channels=24, context_channels=C, downsample=16, grid_config=grid,
                       depth_channels=8, mid_channels=32, use_dcn=False),
        forward_projection=dict(type='LSSViewTransformerFunction3D', grid_config=grid, input_size=(64, 96), downsample=16),
        img_bev_encoder_backbone=dict(type='CustomResNet3D', depth=10, block_strides=[1, 2, 2], n_input_channels=C,
                                      block_inplanes=[16, 32, 32], out_indices=(0, 1, 2), norm_cfg=dict(type='SyncBN')),
        img_bev_encoder_neck=dict(type='FPN3D', in_channels=[16, 32, 32], out_channels=64, norm_cfg=dict(type='SyncBN')),
        occupancy_head=dict(type='OccHead', norm_cfg=dict(type='SyncBN'), soft_weights=True, final_occ_size=[16, 16, 8],
                            empty_idx=18, num_level=3, in_channels=[64] * 3, out_channel=19))
    torch.manual_seed(0)
    m = FBOCC(**cfg, execution=dict(mfma_conv3d=True))
    _randomise(m, 4)
    B = 1
    g = torch.Generator().manual_seed(2)
    bev = torch.randn(B, C, 8, 8, 4, generator=g)
    monkeypatch.setattr(m._path[0], 'forward', lambda cam, ctx, dep, img_metas=None, **kw: bev)
    monkeypatch.setattr(m._path[1], 'fuse_history', lambda x, metas, bda: x)
    pc = S.PathConfig(name='t', input_size=(64, 96), downsample=16, grid_config=grid, channels=C)
    inputs = [torch.randn(B, 6, 3, 64, 96, generator=g)] + list(S.camera_rig(pc, B, seed=0))
    metas = [dict(sequence_group_idx=0, start_of_sequence=True, curr_to_prev_ego_rt=torch.eye(4), index=0)]
    with torch.no_grad():
        ref_raw = m.predict_occupancy(inputs, metas, return_raw_occ=True)         # CPU tensors: module route
        monkeypatch.setattr(_capi, 'conv3d_ndhwc', emu_backend)
        monkeypatch.setattr(_capi, 'blend_levels_ndhwc', emu_blend)
        monkeypatch.setattr(m, '_use_mfma', lambda x: True)
        got_raw = m.predict_occupancy(inputs, metas, return_raw_occ=True)
        got_ids = m.predict_occupancy(inputs, metas)
    assert got_raw.shape == ref_raw.shape == (1, 16, 16, 8, 18)
    assert torch.allclose(got_raw, ref_raw, atol=1e-5, rtol=1e-4), (got_raw - ref_raw).abs().max()
    assert (got_ids == ref_raw.argmax(-1)).float().mean() > 0.999
    m.train()
    assert m._runners is None                                                     # folded weights dropped with the mode


def emu_dgrad(dy, wft, dx, ksize=3, stride=1, pad=1):
    code, y = E.conv3d_dgrad_ndhwc(dy.contiguous(), wft, tuple(dx.shape[1:4]), dx.shape[4], ksize=ksize, stride=stride, pad=pad)
    assert code == 0 and not torch.isnan(y).any()
    return y


def emu_wgrad(x, dy, dw, ksize=3, stride=1, pad=1):
    code, y = E.conv3d_wgrad_ndhwc(x.contiguous(), dy.contiguous(), ksize=ksize, stride=stride, pad=pad)
    assert code == 0 and tuple(y.shape) == tuple(dw.shape)
    return y


def test_training_route_gradients_equal_torch_autograd():
    """CustomResNet3D -> FPN3D -> OccHead in TRAIN mode (batch-statistics BN between the convolutions stays torch):
    forward, input gradient and every parameter gradient through the MFMA autograd route == the nn.Conv3d route."""
    import copy
    from fb_bev_amd.bev_encoder import CustomResNet3D, FPN3D
    from fb_bev_amd.occ_head import OccHead
    torch.manual_seed(0)
    chans = [16, 32, 32]
    net = nn.ModuleDict(dict(
        bb=CustomResNet3D(depth=10, block_strides=[1, 2, 2], n_input_channels=16, block_inplanes=chans, out_indices=(0, 1, 2),
                          norm_cfg=dict(type='SyncBN')),
        neck=FPN3D(in_channels=chans, out_channels=64, norm_cfg=dict(type='SyncBN')),
        head=OccHead(in_channels=[64] * 3, out_channel=19, num_level=3, soft_weights=True, use_focal_loss=False,
                     norm_cfg=dict(type='SyncBN'), final_occ_size=[16, 16, 8], empty_idx=18))).train()
    ref = copy.deepcopy(net)
    n = M.enable_training_route(net, True, backends=(emu_backend, emu_dgrad, emu_wgrad))
    assert n == sum(isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)) for m in net.modules())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 8, 8, 4, generator=g)
    wgt = torch.randn(2, 19, 16, 16, 8, generator=g)
    outs = []
    for mod in (net, ref):
        xi = x.clone().requires_grad_()
        o = mod['head'](mod['neck'](mod['bb'](xi)))['output_voxels'][0]
        (o * wgt).sum().backward()
        outs.append((o.detach(), xi.grad))
    assert torch.allclose(outs[0][0], outs[1][0], atol=2e-4, rtol=1e-4)
    scale = outs[1][1].abs().max()
    assert (outs[0][1] - outs[1][1]).abs().max() <= 2e-4 * scale
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and q.grad is not None, name
        # absolute floor: a conv bias in front of a batch-statistics BN has an analytically zero gradient (rounding noise)
        tol = 3e-4 * q.grad.abs().max() + 2e-4
        assert (p.grad - q.grad).abs().max() <= tol, (name, float((p.grad - q.grad).abs().max()), float(tol))
    # the two tiny output convolutions (19 / 4 channels) are not multiples of 16: MConv3d zero-pads them onto the same route
    assert M._supported_train(net['head'].occ_pred_conv[3], x) and M._supported_train(net['head'].occ_pred_conv[0], x)
    M.enable_training_route(net, False)
    assert not any(getattr(m, 'mfma', False) for m in net.modules())


@pytest.mark.parametrize('B,hw,Cin,Cout,k,s,p,relu,res', [(2, (9, 7), 16, 32, 3, 1, 1, True, True), (1, (8, 11), 32, 64, 3, 2, 1, True, False),
                                                          (1, (5, 6), 64, 16, 1, 1, 0, False, True), (2, (6, 6), 16, 48, 1, 2, 0, False, False)])
def test_conv2d_nhwc_vs_torch(B, hw, Cin, Cout, k, s, p, relu, res):
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(B, Cin, *hw, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    exp = F.conv2d(x, w, b, stride=s, padding=p)
    r = torch.randn(exp.shape, generator=g) if res else None
    exp = exp + r if res else exp
    exp = exp.relu() if relu else exp
    code, y = E.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous(), M.weight_fragments(w[:, :, None]),
                            F.pad(b, (0, (Cout + 15) // 16 * 16 - Cout)), Cout, ksize=k, stride=s, pad=p, relu=relu,
                            residual=None if r is None else r.permute(0, 2, 3, 1).contiguous())
    assert code == 0 and not torch.isnan(y).any()
    assert torch.allclose(y.permute(0, 3, 1, 2), exp, atol=1e-4, rtol=1e-4), (y.permute(0, 3, 1, 2) - exp).abs().max()


@pytest.mark.parametrize('depth', [18, 50])
def test_image_encoder_runners_equal_the_modules(depth):
    from fb_bev_amd.img_encoder import CustomFPN, ResNet
    torch.manual_seed(depth)
    net = ResNet(depth=depth, base_channels=16, num_stages=4, out_indices=(2, 3), norm_eval=False)
    exp = 1 if depth == 18 else 4
    neck = CustomFPN(in_channels=[64 * exp, 128 * exp], out_channels=32, num_outs=1, start_level=0, out_ids=[0])
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in list(net.modules()) + list(neck.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.4 - 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.4 + 0.3)      # < 1: keeps 16 blocks well scaled
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.2 - 0.1)
    net.eval(); neck.eval()
    img = torch.randn(2, 3, 64, 96, generator=g)
    with torch.no_grad():
        ref_feats = net(img)
        ref = neck(ref_feats)
        feats = M.ResNetRunner(net)(img, backend=emu_backend)
        got = M.CustomFPNRunner(neck)(feats, backend=emu_backend)
    for a, b in zip(feats, ref_feats):
        assert torch.allclose(a.permute(0, 3, 1, 2), b, atol=1e-4, rtol=1e-4), (a.permute(0, 3, 1, 2) - b).abs().max()
    assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-4, rtol=1e-4)


def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize('B,dims,Cin,Cout,k,s,p,relu,res,planar', [
    (1, (5, 6, 3), 32, 16, 3, 1, 1, False, False, False),
    (2, (5, 6, 3), 64, 64, 3, 2, 1, True, True, False),
    (1, (3, 5, 2), 96, 19, 1, 1, 0, False, False, False),          # odd number of 32-channel groups; scalar stores
    (1, (1, 9, 7), 32, 32, 3, 1, 1, True, False, True),            # planar: the 2-D case
    (1, (1, 8, 6), 64, 48, 1, 2, 0, False, True, True),
])
def test_conv3d_bf16_kernel_vs_torch_on_rounded_operands(B, dims, Cin, Cout, k, s, p, relu, res, planar):
    """fbbev_conv3d_ndhwc_bf16 == an fp32 convolution of the bf16-rounded input and weight (fp32 accumulation)."""
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(B, Cin, *dims, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    if planar:
        w = w[:, :, :1].contiguous()
    b = torch.randn(Cout, generator=g)
    pad = (0, p, p) if planar else p
    exp = F.conv3d(_bf(x), _bf(w), b, stride=(1, s, s) if planar else s, padding=pad)
    r = torch.randn(exp.shape, generator=g) if res else None
    exp = exp + r if res else exp
    exp = exp.relu() if relu else exp
    code, y = E.conv3d_ndhwc_bf16(M.to_ndhwc(x), M.weight_fragments_bf16(w), F.pad(b, (0, (Cout + 15) // 16 * 16 - Cout)), Cout, ksize=k,
                                  stride=s, pad=p, relu=relu, residual=None if r is None else M.to_ndhwc(r), planar=planar)
    assert code == 0 and not torch.isnan(y).any()
    assert torch.allclose(M.to_ncdhw(y), exp, atol=1e-4, rtol=1e-4), (M.to_ncdhw(y) - exp).abs().max()
    # and it is a bf16-precision approximation of the fp32 convolution
    full = F.conv3d(x, w, b, stride=(1, s, s) if planar else s, padding=pad)
    full = (full + r if res else full)
    full = full.relu() if relu else full
    assert (M.to_ncdhw(y) - full).abs().max() < 3e-2


def test_transposed_conv_bf16_vs_torch():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 3, 4, 2, generator=g)
    w = torch.randn(64, 24, 2, 2, 2, generator=g) / 8
    exp = F.conv_transpose3d(_bf(x), _bf(w), None, stride=2)
    code, y = E.conv3d_ndhwc_bf16(M.to_ndhwc(x), M.weight_fragments_bf16(w, transposed=True), torch.zeros(32), 24, transposed=True)
    assert code == 0 and torch.allclose(M.to_ncdhw(y), exp, atol=1e-4, rtol=1e-4)
    assert E.conv3d_ndhwc_bf16(torch.zeros(1, 2, 2, 2, 16), torch.zeros(27 * 512, dtype=torch.bfloat16), torch.zeros(16), 16)[0] == -2


def test_bf16_runners_track_the_fp32_modules():
    """precision='bf16': layers with Cin % 32 == 0 take the bf16-MFMA kernel (fp32 activations in and out), the others the fp32
    one; the stack output stays within bf16 accuracy of the fp32 modules."""
    from fb_bev_amd.bev_encoder import CustomResNet3D, FPN3D
    from fb_bev_amd.occ_head import OccHead
    torch.manual_seed(0)
    chans = [32, 32, 64]
    bb = _randomise(CustomResNet3D(depth=10, block_strides=[1, 2, 2], n_input_channels=16, block_inplanes=chans,
                                   out_indices=(0, 1, 2), norm_cfg=dict(type='SyncBN')), 1)
    neck = _randomise(FPN3D(in_channels=chans, out_channels=64, norm_cfg=dict(type='SyncBN')), 2)
    head = _randomise(OccHead(in_channels=[64] * 3, out_channel=19, num_level=3, soft_weights=True, use_focal_loss=False,
                              norm_cfg=dict(type='SyncBN'), final_occ_size=[16, 16, 8], empty_idx=18), 3)
    rb, rn, rh = M.ResNet3DRunner(bb, 'bf16_tiled'), M.FPN3DRunner(neck, 'bf16_tiled'), M.OccHeadRunner(head, 'bf16_tiled')
    assert rb.stages[0][0][1].tiled and rn.outs[0].tiled and not rh.deblock.tiled and not rb.stages[1][0][0].tiled   # stride 2: direct
    assert rb.input_proj.wf.dtype == torch.float32                          # 16 input channels: fp32 kernel
    assert rb.stages[0][0][1].wf.dtype == torch.bfloat16 and rn.outs[0].wf.dtype == torch.bfloat16
    assert rh.deblock.wf.dtype == torch.bfloat16 and rh.pred[1].wf.dtype == torch.float32      # 16 -> 19: fp32
    x = torch.randn(1, 16, 8, 8, 4)
    with torch.no_grad():
        ref = head(neck(bb(x)))['output_voxels'][0]
        got = rh(rn(rb(M.to_ndhwc(x), backend=emu_backend), backend=emu_backend), backend=emu_backend, blend_backend=emu_blend)
    err = (got - ref).abs().max() / ref.abs().max()
    assert 0 < err < 3e-2, err


@pytest.mark.parametrize('B,dims,Cin,Cout,relu,res', [
    (1, (4, 8, 8), 32, 16, False, False),           # exactly one tile
    (2, (5, 11, 9), 64, 64, True, True),            # partial tiles along all three axes, two samples, MT=4
    (1, (8, 3, 8), 32, 19, False, False),           # scalar stores, a tile thinner than its height
])
def test_conv3d_tiled_bf16_kernel_equals_direct_bf16_kernel_and_torch(B, dims, Cin, Cout, relu, res):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, *dims, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=g)
    exp = F.conv3d(_bf(x), _bf(w), b, padding=1)
    r = torch.randn(exp.shape, generator=g) if res else None
    exp = exp + r if res else exp
    exp = exp.relu() if relu else exp
    wfb = M.weight_fragments_bf16(w)
    bias = F.pad(b, (0, (Cout + 15) // 16 * 16 - Cout))
    rn = None if r is None else M.to_ndhwc(r)
    code, y = E.conv3d_k3s1_tiled_bf16(M.to_ndhwc(x), wfb, bias, Cout, relu=relu, residual=rn)
    assert code == 0 and not torch.isnan(y).any()
    assert torch.allclose(M.to_ncdhw(y), exp, atol=1e-4, rtol=1e-4), (M.to_ncdhw(y) - exp).abs().max()
    code, y2 = E.conv3d_ndhwc_bf16(M.to_ndhwc(x), wfb, bias, Cout, relu=relu, residual=rn)
    assert code == 0 and torch.allclose(y, y2, atol=2e-5, rtol=1e-5)
