"""FBOCC detector assembly: the shipped config's `model` block builds unchanged, parameter names equal
the reference's (fixture: names / shapes of the REAL in-tree reference blocks at the shipped config), and the host-side
plumbing of forward_train / simple_test runs on CPU with stand-ins for the two GPU-only stages."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _model_cfg(name='fbocc-r50-cbgs_depth_16f_16x4_20e.py'):
    return json.load(open(os.path.join(GOLD, 'fbocc_config_path_blocks.json')))[name]['model']


@pytest.fixture(scope='module')
def shipped():
    from fb_bev_amd.fbocc import FBOCC
    cfg = dict(_model_cfg())
    assert cfg.pop('type') == 'FBOCC'
    torch.manual_seed(0)
    return FBOCC(**cfg), cfg


def test_shipped_config_builds_and_parameter_names_equal_the_reference(shipped):
    model, cfg = shipped
    sd = model.state_dict()
    ref = json.load(open(os.path.join(GOLD, 'fbocc_reference_state_keys.json')))
    for block, keys in ref.items():
        mine = {k[len(block) + 1:]: list(v.shape) for k, v in sd.items() if k.startswith(block + '.')}
        assert mine == keys, (block, sorted(set(mine) ^ set(keys))[:5])
    # the two fusion convolutions live on the detector itself (fbocc.py:111-127), once
    assert sd['history_keyframe_time_conv.0.weight'].shape == (80, 81, 1, 1, 1)
    assert sd['history_keyframe_cat_conv.0.weight'].shape == (80, 80 * 17, 1, 1, 1)
    assert not any(k.startswith('_path') for k in sd)
    tops = {k.split('.')[0] for k in sd}
    # forward_projection.{dx,bx,nx} are frozen nn.Parameters in the reference as well (view_transformer.py:356-358)
    assert [k for k in sd if k.startswith('forward_projection.')] == ['forward_projection.' + n for n in ('dx', 'bx', 'nx')]
    assert tops == {'img_backbone', 'img_neck', 'depth_net', 'forward_projection', 'backward_projection', 'history_keyframe_time_conv',
                    'history_keyframe_cat_conv', 'img_bev_encoder_backbone', 'img_bev_encoder_neck', 'occupancy_head'}
    assert model.fix_void and model.readd and model.use_depth_supervision
    assert model.do_history == cfg['do_history']
    model.train()
    assert model.history.training and model.view_transform.training
    model.eval()
    assert not model.history.training


def test_trt_config_block_builds_too():
    from fb_bev_amd.fbocc import FBOCC
    cfg = dict(_model_cfg('fbocc-r50-cbgs_depth_16f_16x4_20e_trt.py'))
    cfg.pop('type')                       # FBOCCTRT: same constructor blocks (fbocc_trt.py subclasses FBOCC)
    m = FBOCC(**cfg)
    assert m.occupancy_head.out_channel == 19


def test_unused_reference_branches_are_rejected():
    from fb_bev_amd.fbocc import FBOCC
    cfg = dict(_model_cfg())
    cfg.pop('type')
    with pytest.raises(NotImplementedError):
        FBOCC(**{**cfg, 'frpn': dict(type='FRPN')})
    with pytest.raises(KeyError):
        FBOCC(**{**cfg, 'img_neck': dict(type='SomethingElse')})


def test_cvpr2023_axis_convention_equals_reference_sequence():
    """predict_occupancy takes the argmax first and shuffles one id per voxel; the reference (fbocc.py:540-557) shuffles
    the probabilities and takes the argmax last."""
    g = torch.Generator().manual_seed(0)
    occ = torch.randn(1, 19, 6, 5, 4, generator=g)
    p = occ.permute(0, 2, 3, 4, 1)[0][..., 1:].softmax(-1)                # reference op sequence
    p = p.permute(3, 2, 0, 1)
    p = torch.flip(p, [2])
    p = torch.rot90(p, -1, [2, 3])
    p = p.permute(2, 3, 1, 0)
    exp_raw, exp = p, p.argmax(-1)
    x = occ[:, 1:].softmax(1)
    for raw in (False, True):
        y = x if raw else x.argmax(1, keepdim=True)
        y = y.permute(0, 1, 4, 2, 3)
        y = torch.rot90(torch.flip(y, [3]), -1, [3, 4]).permute(0, 3, 4, 2, 1)
        if raw:
            assert torch.allclose(y[0], exp_raw, atol=1e-7, rtol=1e-6)      # softmax over a strided vs a contiguous axis
        else:
            assert torch.equal(y[0, ..., 0], exp)


def test_forward_train_and_simple_test_plumbing_on_cpu(monkeypatch):
    """Small detector; the two GPU-only stages (view transformation, history fusion) are replaced by shape-correct CPU
    stand-ins so that the host logic around them -- encoders, head, loss dict, prediction format -- is exercised."""
    from fb_bev_amd.fbocc import FBOCC
    from fb_bev_amd import synthetic as S
    grid = {'x': [-8, 8, 0.8], 'y': [-8, 8, 0.8], 'z': [-1, 2.2, 0.8], 'depth': [2.0, 10.0, 1.0]}     # 20x20x4, D=8
    pcr = [-8, -8, -1, 8, 8, 2.2]
    C = 16
    cfg = dict(
        use_depth_supervision=True, fix_void=True, do_history=True, history_cat_num=2, single_bev_num_channels=C, readd=True,
        img_backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(2, 3), norm_eval=False, base_channels=8),
        img_neck=dict(type='CustomFPN', in_channels=[32, 64], out_channels=24, num_outs=1, start_level=0, out_ids=[0]),
        depth_net=dict(type='CM_DepthNet', in_channels=24, context_channels=C, downsample=16, grid_config=grid,
                       depth_channels=8, mid_channels=32, loss_depth_weight=1., use_dcn=False),
        forward_projection=dict(type='LSSViewTransformerFunction3D', grid_config=grid, input_size=(64, 96), downsample=16),
        backward_projection=None,
        img_bev_encoder_backbone=dict(type='CustomResNet3D', depth=18, block_strides=[1, 2, 2], n_input_channels=C,
                                      block_inplanes=[8, 16, 32], out_indices=(0, 1, 2), norm_cfg=dict(type='SyncBN')),
        img_bev_encoder_neck=dict(type='FPN3D', in_channels=[8, 16, 32], out_channels=16, norm_cfg=dict(type='SyncBN')),
        occupancy_head=dict(type='OccHead', use_focal_loss=True, norm_cfg=dict(type='SyncBN'), soft_weights=True,
                            final_occ_size=[40, 40, 8], empty_idx=18, num_level=3, in_channels=[16] * 3, out_channel=19,
                            point_cloud_range=pcr))
    torch.manual_seed(0)
    m = FBOCC(**cfg)
    B, N = 2, 6
    calls = {}

    def vt_stub(cam_params, context, depth, img_metas=None, **kw):
        calls['vt'] = (tuple(context.shape), tuple(depth.shape))
        assert context.dtype == depth.dtype == torch.float32
        pooled = (context.mean((1, 3, 4))[:, :, None, None, None] + depth.mean((1, 2, 3, 4)).view(-1, 1, 1, 1, 1))
        return pooled.expand(B, C, 20, 20, 4) + torch.linspace(0, 1, 20).view(1, 1, 20, 1, 1)

    def hist_stub(bev, img_metas, bda):
        calls['hist'] = (tuple(bev.shape), tuple(bda.shape), len(img_metas))
        return bev
    monkeypatch.setattr(m._path[0], 'forward', vt_stub)
    monkeypatch.setattr(m._path[1], 'fuse_history', hist_stub)

    pc = S.PathConfig(name='t', input_size=(64, 96), downsample=16, grid_config=grid, channels=C)
    cam = S.camera_rig(pc, B, seed=0, bda_aug=True)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(B, N, 3, 64, 96, generator=g)
    metas = [dict(sequence_group_idx=b, start_of_sequence=True, curr_to_prev_ego_rt=torch.eye(4), index=b) for b in range(B)]
    gt_occ = torch.randint(1, 19, (B, 40, 40, 8), generator=g)
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.3] = 255
    gt_depth = torch.rand(B, N, 64, 96, generator=g) * 9 + 2
    gt_depth[torch.rand(gt_depth.shape, generator=g) < 0.9] = 0

    m.train()
    losses = m(return_loss=True, img_inputs=[img] + list(cam), img_metas=metas, gt_occupancy=gt_occ, gt_depth=gt_depth)
    assert set(losses) == {'loss_voxel_ce_c_0', 'loss_voxel_sem_scal_c_0', 'loss_voxel_geo_scal_c_0', 'loss_voxel_lovasz_c_0',
                           'loss_depth'}
    assert calls['vt'] == ((B, N, C, 4, 6), (B, N, 8, 4, 6)) and calls['hist'] == ((B, C, 20, 20, 4), (B, 3, 3), B)
    total = m.parse_losses(losses)
    total.backward()
    assert torch.isfinite(total)
    for name in ('img_backbone.conv1.weight', 'img_neck.lateral_convs.0.conv.weight', 'depth_net.context_conv.weight',
                 'img_bev_encoder_backbone.input_proj.0.weight', 'occupancy_head.occ_pred_conv.3.weight'):
        p = dict(m.named_parameters())[name]
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, name

    m.eval()
    with torch.no_grad():
        ids = m.predict_occupancy([img] + list(cam), metas)
        assert ids.shape == (B, 40, 40, 8) and ids.dtype == torch.int64 and int(ids.max()) <= 17
        one = [t[:1] for t in [img] + list(cam)]
        res = m(return_loss=False, img_inputs=[one], img_metas=[metas[:1]])
    assert isinstance(res, list) and len(res) == 1 and res[0]['pred_occupancy'].shape == (40, 40, 8)
    assert res[0]['index'] == 0 and isinstance(res[0]['pred_occupancy'], np.ndarray)
    assert m.do_history is True
    with pytest.raises(TypeError):
        m(return_loss=False, img_inputs=tuple(one), img_metas=[metas[:1]])
    with pytest.raises(ValueError):
        m(return_loss=False, img_inputs=one, img_metas=[metas[:1]])


@pytest.mark.parametrize('name', ['fbocc-r50-cbgs_depth_16f_16x4_20e.py', 'fbocc-r50-cbgs_depth_16f_16x4_20e_trt.py'])
def test_build_detector_from_config_block(name):
    from fb_bev_amd import config as C
    block = json.load(open(os.path.join(GOLD, 'fbocc_config_path_blocks.json')))[name]['model']
    m = C.build_detector(block, execution=dict(with_cp=False))
    assert type(m).__name__ == 'FBOCC' and m.img_backbone.with_cp is False and m.occupancy_head.with_cp is False
    with pytest.raises(KeyError):
        C.build_detector({**block, 'type': 'BEVDet'})


def test_detector_fails_loudly_without_a_gpu(shipped):
    """No CPU fallback: the assembled detector refuses CPU tensors at the first HIP stage instead of computing elsewhere."""
    from fb_bev_amd import _capi, synthetic as S
    model, _ = shipped
    model.eval()
    pc = S.CONFIGS['REF']
    cam = S.camera_rig(pc, 1, seed=0)
    feats = torch.zeros(1, 6, 256, 16, 44)
    with torch.no_grad(), pytest.raises(_capi.FbbevError):
        mlp = model.depth_net.get_mlp_input(*cam)
        context, depth = model.depth_net(feats, mlp)
        model.view_transform(list(cam), context, depth)
