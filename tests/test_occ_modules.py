"""SURVEY 8f-3 blocks against fixtures of the REAL reference classes (tests/golden/make_golden_occ.py):
CustomResNet3D, FPN3D, OccHead (+ focal / CE / sem-scal / geo-scal / Lovasz losses), CustomFPN.
Tolerance: fp32 convolution stacks 1e-4 abs on O(1) activations; scalar losses 2e-5 relative (the restated losses sum
masked values over all voxels instead of compacting them first: fp32 summation order differs)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), 'golden', 'occ_encoder_head_small.npz')


@pytest.fixture(scope='module')
def gold():
    return {k: v for k, v in np.load(G).items()}


def _load(mod, gold, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in gold.items() if k.startswith(prefix)}
    missing, unexpected = mod.load_state_dict(sd, strict=True), None
    return mod.eval()


def test_resnet3d_and_fpn3d_match_reference_fixture(gold):
    from fb_bev_amd.bev_encoder import CustomResNet3D, FPN3D
    chans = [8, 16, 32]
    bb = _load(CustomResNet3D(depth=18, block_strides=[1, 2, 2], n_input_channels=6, block_inplanes=chans, out_indices=(0, 1, 2),
                              norm_cfg=dict(type='BN3d', requires_grad=True)), gold, 'w.backbone.')
    neck = _load(FPN3D(in_channels=chans, out_channels=16, norm_cfg=dict(type='BN3d', requires_grad=True)), gold, 'w.neck.')
    with torch.no_grad():
        feats = bb(torch.from_numpy(gold['vox.x']))
        outs = neck(feats)
    assert len(feats) == len(outs) == 3
    for i in range(3):
        assert torch.allclose(feats[i], torch.from_numpy(gold[f'vox.backbone{i}']), atol=1e-4, rtol=1e-4)
        assert torch.allclose(outs[i], torch.from_numpy(gold[f'vox.neck{i}']), atol=1e-4, rtol=1e-4)
    # shipped config: SyncBN builds the same parameters / names
    bb2 = CustomResNet3D(depth=18, block_strides=[1, 2, 2], n_input_channels=6, block_inplanes=chans, out_indices=(0, 1, 2),
                         norm_cfg=dict(type='SyncBN', requires_grad=True))
    assert list(bb2.state_dict()) == list(bb.state_dict())


def test_resnet3d_depth10_and_groupnorm_neck(gold):
    from fb_bev_amd.bev_encoder import CustomResNet3D, FPN3D
    bb = _load(CustomResNet3D(depth=10, block_strides=[2, 2], n_input_channels=4, block_inplanes=[16, 32], out_indices=(1,),
                              norm_cfg=dict(type='BN3d', requires_grad=True)), gold, 'w.backbone50.')
    neck = _load(FPN3D(in_channels=[32], out_channels=8, norm_cfg=dict(type='GN', num_groups=4, requires_grad=True)), gold,
                 'w.neckg.')
    with torch.no_grad():
        f = bb(torch.from_numpy(gold['vox50.x']))
        n = neck(f)
    assert torch.allclose(f[0], torch.from_numpy(gold['vox50.backbone']), atol=1e-4, rtol=1e-4)
    assert torch.allclose(n[0], torch.from_numpy(gold['vox50.neck']), atol=1e-4, rtol=1e-4)
    with pytest.raises(NotImplementedError):
        CustomResNet3D(depth=50)


def _head(gold, focal):
    from fb_bev_amd.occ_head import OccHead
    h = OccHead(in_channels=[8, 8, 8], out_channel=19, num_level=3, soft_weights=True, use_focal_loss=focal,
                norm_cfg=dict(type='BN3d', requires_grad=True), final_occ_size=[200, 200, 2], empty_idx=18,
                loss_weight_cfg=dict(loss_voxel_ce_weight=1.0, loss_voxel_sem_scal_weight=0.7, loss_voxel_geo_scal_weight=1.3,
                                     loss_voxel_lovasz_weight=0.9))
    return _load(h, gold, 'w.head.')


def test_occ_head_forward_and_losses_match_reference_fixture(gold):
    h = _head(gold, True)
    assert torch.allclose(h.class_weights, torch.from_numpy(gold['head.class_weights']).float(), rtol=1e-6)
    vf = [torch.from_numpy(gold[f'head.feat{i}']) for i in range(3)]
    gt = torch.from_numpy(gold['head.gt'].astype(np.int64))
    with torch.no_grad():
        logits = h(vf)['output_voxels'][0]
    assert logits.shape == (1, 19, 200, 200, 2)
    assert torch.allclose(logits[:, :, ::5, ::5], torch.from_numpy(gold['head.logits_s5']), atol=1e-4, rtol=1e-4)
    assert abs(float(logits.double().sum()) - float(gold['head.logits_sum'])) < 1e-4 * logits.numel() ** 0.5 + 1.0
    for tag, head in (('focal', h), ('ce', _head(gold, False))):
        with torch.no_grad():
            losses = head.loss(output_voxels=[logits.clone()], target_voxels=gt)
        assert set(losses) == {f'loss_voxel_{n}_c_0' for n in ('ce', 'sem_scal', 'geo_scal', 'lovasz')}
        for k, v in losses.items():
            exp = float(gold[f'head.{tag}.{k}'])
            assert abs(float(v) - exp) <= 2e-5 * abs(exp) + 1e-6, (tag, k, float(v), exp)


def test_occ_losses_have_gradients_and_no_python_branches_on_data(gold):
    """The restated losses are differentiable end to end and contain no data-dependent host branch: running them under
    torch's sync debug mode is a GPU-only check (tests/test_gpu_full_model.py); here the structure is exercised with a
    target that lacks several classes and has no ignored voxel at all."""
    from fb_bev_amd import occ_loss as L
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 19, 6, 5, 4, generator=g, requires_grad=True)
    gt = torch.randint(14, 19, (2, 6, 5, 4), generator=g)
    total = (L.sem_scal_loss(logits, gt) + L.geo_scal_loss(logits, gt, non_empty_idx=18) +
             L.lovasz_softmax(torch.softmax(logits, 1), gt, ignore=255) + L.CE_ssc_loss(logits, gt, L.class_weights(19).float()))
    total.backward()
    assert torch.isfinite(total) and torch.isfinite(logits.grad).all() and logits.grad.abs().sum() > 0


def test_gt_majority_vote_resize_matches_reference_fixture(gold):
    from fb_bev_amd.occ_head import OccHead
    h = OccHead(in_channels=[8], out_channel=19, num_level=1, soft_weights=False, use_focal_loss=False,
                norm_cfg=dict(type='BN3d', requires_grad=True), final_occ_size=[8, 8, 4], empty_idx=18, use_deblock=False)
    with torch.no_grad():
        losses = h.loss(output_voxels=[torch.from_numpy(gold['resize.logits'])], target_voxels=torch.from_numpy(gold['resize.gt']))
    for k, v in losses.items():
        exp = float(gold[f'resize.{k}'])
        assert abs(float(v) - exp) <= 2e-5 * abs(exp) + 1e-6, (k, float(v), exp)


def test_custom_fpn_matches_reference_fixture(gold):
    from fb_bev_amd.img_encoder import CustomFPN
    cf = _load(CustomFPN(in_channels=[12, 24], out_channels=8, num_outs=1, start_level=0, out_ids=[0]), gold, 'w.fpn.')
    with torch.no_grad():
        y = cf([torch.from_numpy(gold['fpn.c4']), torch.from_numpy(gold['fpn.c5'])])
    assert torch.allclose(y, torch.from_numpy(gold['fpn.out']), atol=1e-5, rtol=1e-5)


def test_resnet50_structure_matches_published_checkpoint_layout():
    """State-dict names / shapes of torchvision's resnet50 checkpoint (`resnet50-0676ba61.pth`, named by the config)."""
    from fb_bev_amd.img_encoder import ResNet
    net = ResNet(depth=50, num_stages=4, out_indices=(2, 3), norm_eval=False, style='pytorch')
    sd = net.state_dict()
    assert sd['conv1.weight'].shape == (64, 3, 7, 7)
    assert sd['layer1.0.downsample.0.weight'].shape == (256, 64, 1, 1)
    assert sd['layer2.0.conv2.weight'].shape == (128, 128, 3, 3)
    assert sd['layer4.2.conv3.weight'].shape == (2048, 512, 1, 1)
    assert sum(p.numel() for p in net.parameters()) == 23508032          # resnet50 without the fc layer
    assert [len(getattr(net, f'layer{i}')) for i in (1, 2, 3, 4)] == [3, 4, 6, 3]
    net.eval()
    with torch.no_grad():
        c4, c5 = net(torch.randn(1, 3, 64, 96))
    assert c4.shape == (1, 1024, 4, 6) and c5.shape == (1, 2048, 2, 3)
    assert net.layer2[0].conv2.stride == (2, 2) and net.layer2[0].conv1.stride == (1, 1)     # 'pytorch' style


def test_lovasz_blockwise_prefix_sum_is_exact():
    """occ_loss._cumsum_rows (two-level scan of the Lovasz prefix sums) and the single-prefix-sum form of the union:
    identical values to ATen's cumsum on 0/1 data, and the loss of a long input equals the two-cumsum formulation of
    lovasz_softmax.py:20-33 bit for bit."""
    import torch
    from fb_bev_amd import occ_loss as OL
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(5, 100003, generator=g) < 0.3).float()
    assert torch.equal(OL._cumsum_rows(x), x.cumsum(1))
    assert torch.equal(OL._cumsum_rows(x[:, :100]), x[:, :100].cumsum(1))
    C, N = 4, 40000
    probas = torch.rand(1, C, N, 1, 1, generator=g).softmax(1)
    labels = torch.randint(0, C + 1, (1, N, 1, 1), generator=g)
    labels[labels == C] = 255
    got = OL.lovasz_softmax(probas, labels, ignore=255)
    # reference formulation, per class (lovasz_softmax.py:20-33,139-160 with classes='present')
    p = probas.reshape(C, N); lab = labels.reshape(-1); valid = lab != 255
    losses = []
    for c in range(C):
        fg = (lab[valid] == c).float()
        if fg.sum() == 0:
            continue
        err = (fg - p[c][valid]).abs()
        es, perm = torch.sort(err, 0, descending=True)
        fs = fg[perm]
        gts = fs.sum()
        jac = 1.0 - (gts - fs.cumsum(0)) / (gts + (1 - fs).cumsum(0))
        jac[1:] = jac[1:] - jac[:-1].clone()
        losses.append(torch.dot(es, jac))
    exp = torch.stack(losses).mean()
    assert abs(float(got) - float(exp)) <= 2e-6 * abs(float(exp))
