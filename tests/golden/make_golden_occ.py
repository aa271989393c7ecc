#!/usr/bin/env python3
"""Fixtures for the voxel encoder, occupancy head and occupancy losses: run the REAL reference modules
    CustomResNet3D  mmdet3d/models/fbbev/modules/resnet3d.py
    FPN3D           mmdet3d/models/fbbev/modules/fpn3d.py
    OccHead         mmdet3d/models/fbbev/heads/occupancy_head.py  (+ occ_loss_utils/{lovasz_softmax,semkitti,focal_loss,
                    nusc_param}.py)
    CustomFPN       mmdet3d/models/necks/fpn.py
on CPU at small sizes and store inputs, state dicts and outputs (forward and the four loss terms).

The files are loaded by path.  mmcv / mmdet / spconv are absent, so stand-ins are installed for the external
constructors the files call: mmcv.cnn.build_conv_layer / build_norm_layer / ConvModule map the config dicts onto the
torch layers mmcv itself would build (Conv3d, ConvTranspose3d, BatchNorm3d, GroupNorm; ConvModule = conv -> norm -> ReLU
with children `conv`, `bn`/`gn`, `activate`).  No arithmetic of the reference files themselves is replaced.  The
reference's CustomFocalLoss moves a constant to the GPU in its constructor (`.cuda()`, focal_loss.py:230) -- patched to
the identity for this CPU run.

Run in the build container:  python tests/golden/make_golden_occ.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402


def _norm(cfg, n, dims=3):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    cfg.pop('requires_grad', None)
    if typ == 'GN':
        return 'gn', nn.GroupNorm(num_channels=n, **cfg)
    if typ == 'SyncBN':                      # mmcv builds nn.SyncBatchNorm: same parameters / state names as BatchNorm
        return 'bn', {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dims](n, **cfg)
    return 'bn', {'BN3d': nn.BatchNorm3d, 'BN': nn.BatchNorm2d, 'BN2d': nn.BatchNorm2d}[typ](n, **cfg)


def _conv(cfg, *a, **k):
    cfg = dict(cfg or dict(type='Conv2d'))
    typ = cfg.pop('type')
    k.update(cfg)
    cls = {'Conv3d': nn.Conv3d, 'deconv3d': nn.ConvTranspose3d, 'Conv2d': nn.Conv2d}[typ]
    if a:
        return cls(*a, **k)
    return cls(k.pop('in_channels'), k.pop('out_channels'), k.pop('kernel_size'), **k)


class _ConvModule(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 bias='auto', inplace=True):
        super().__init__()
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = _conv(conv_cfg, cin, cout, kernel_size, stride=stride, padding=padding, bias=bias)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = _norm(norm_cfg, cout, 3 if isinstance(self.conv, nn.Conv3d) else 2)
            self.add_module(self.norm_name, norm)
        self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

    def forward(self, x):
        x = self.conv(x)
        if self.norm_name:
            x = getattr(self, self.norm_name)(x)
        return self.activate(x) if self.activate is not None else x


def install():
    MG.install_stubs()
    cnn = sys.modules['mmcv.cnn']
    cnn.build_conv_layer, cnn.build_norm_layer, cnn.build_upsample_layer, cnn.ConvModule = _conv, _norm, None, _ConvModule
    sp = MG._mod('spconv'); sp.pytorch = MG._mod('spconv.pytorch', functional=MG._mod('spconv.pytorch.functional'))
    sys.modules['mmdet3d.models.builder'].BACKBONES = MG._Registry()
    sys.modules['mmdet.models'].NECKS = MG._Registry()
    sys.modules['mmdet.models'].HEADS = MG._Registry()
    MG._mod('mmdet.core', reduce_mean=None)
    MG._mod('mmcv.ops', sigmoid_focal_loss=None)
    MG._mod('mmdet.models.builder', LOSSES=MG._Registry())
    MG._mod('mmdet.models.losses'); MG._mod('mmdet.models.losses.utils', weight_reduce_loss=None)
    torch.Tensor.cuda = lambda self, *a, **k: self                      # focal_loss.py:230
    pk = 'mmdet3d.models.fbbev.modules.occ_loss_utils'
    MG._mod('mmdet3d.models.fbbev.modules')
    loss_pkg = MG._mod(pk)
    for f in ('lovasz_softmax', 'nusc_param', 'semkitti', 'focal_loss'):
        m = MG.load_ref(f'{pk}.{f}', f'mmdet3d/models/fbbev/modules/occ_loss_utils/{f}.py')
        for k, v in vars(m).items():
            if not k.startswith('_'):
                setattr(loss_pkg, k, v)
    builder = sys.modules['mmdet3d.models.builder']
    builder.build_loss = lambda cfg: loss_pkg.CustomFocalLoss(**{k: v for k, v in cfg.items() if k != 'type'})
    sys.modules['mmdet3d.models'].builder = builder
    return loss_pkg


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.6 - 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            if isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d, nn.GroupNorm)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.6 + 0.7)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.4 - 0.2)
            if isinstance(m, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose3d)) and m.bias is not None:
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.2 - 0.1)


def main():
    install()
    out = {}
    torch.manual_seed(0)
    r3d = MG.load_ref('refmod.resnet3d', 'mmdet3d/models/fbbev/modules/resnet3d.py')
    fpn3d = MG.load_ref('refmod.fpn3d', 'mmdet3d/models/fbbev/modules/fpn3d.py')
    head = MG.load_ref('refmod.occupancy_head', 'mmdet3d/models/fbbev/heads/occupancy_head.py')
    MG._mod('mmdet3d.models.necks').__path__ = []
    sys.modules['mmdet3d.models'].__path__ = []
    fpn = MG.load_ref('mmdet3d.models.necks.fpn', 'mmdet3d/models/necks/fpn.py')

    # ---- voxel backbone + neck (shipped structure: depth 18, 3 stages, strides 1/2/2, BN; small widths)
    chans = [8, 16, 32]
    bb = r3d.CustomResNet3D(depth=18, block_strides=[1, 2, 2], n_input_channels=6, block_inplanes=chans,
                            out_indices=(0, 1, 2), norm_cfg=dict(type='BN3d', requires_grad=True))
    neck = fpn3d.FPN3D(in_channels=chans, out_channels=16, norm_cfg=dict(type='BN3d', requires_grad=True))
    _randomise(bb, 1); _randomise(neck, 2)
    bb.eval(); neck.eval()
    x = torch.randn(2, 6, 12, 8, 4)
    with torch.no_grad():
        feats = bb(x)
        nfeats = neck(feats)
    out['vox.x'] = x.numpy()
    for i, (a, b) in enumerate(zip(feats, nfeats)):
        out[f'vox.backbone{i}'] = a.numpy(); out[f'vox.neck{i}'] = b.numpy()
    for k, v in bb.state_dict().items():
        out['w.backbone.' + k] = v.numpy()
    for k, v in neck.state_dict().items():
        out['w.neck.' + k] = v.numpy()
    # depth 10 (one block per stage) and a GroupNorm neck: the other code paths of the two files.  (depth 50/101 cannot be
    # built in the reference: _make_layer passes use_spase_3dtensor to Bottleneck, which does not take it -> TypeError.)
    bb50 = r3d.CustomResNet3D(depth=10, block_strides=[2, 2], n_input_channels=4, block_inplanes=[16, 32], out_indices=(1,),
                              norm_cfg=dict(type='BN3d', requires_grad=True))
    neckg = fpn3d.FPN3D(in_channels=[32], out_channels=8, norm_cfg=dict(type='GN', num_groups=4, requires_grad=True))
    _randomise(bb50, 3); _randomise(neckg, 4)
    bb50.eval(); neckg.eval()
    x50 = torch.randn(1, 4, 6, 6, 4)
    with torch.no_grad():
        f50 = bb50(x50)
        n50 = neckg(f50)
    out['vox50.x'], out['vox50.backbone'], out['vox50.neck'] = x50.numpy(), f50[0].numpy(), n50[0].numpy()
    for k, v in bb50.state_dict().items():
        out['w.backbone50.' + k] = v.numpy()
    for k, v in neckg.state_dict().items():
        out['w.neckg.' + k] = v.numpy()

    # ---- occupancy head: forward + the four loss terms (focal variant and CE variant), 19 classes, fix_void layout
    H, W, D = 200, 200, 2            # CustomFocalLoss hard-codes a 200x200 BEV plane (focal_loss.py:225)
    for tag, focal in (('focal', True), ('ce', False)):
        h = head.OccHead(in_channels=[8, 8, 8], out_channel=19, num_level=3, soft_weights=True, use_focal_loss=focal,
                         norm_cfg=dict(type='BN3d', requires_grad=True), final_occ_size=[H, W, D], empty_idx=18,
                         loss_weight_cfg=dict(loss_voxel_ce_weight=1.0, loss_voxel_sem_scal_weight=0.7,
                                              loss_voxel_geo_scal_weight=1.3, loss_voxel_lovasz_weight=0.9))
        _randomise(h, 5)
        h.eval()
        g = torch.Generator().manual_seed(6)
        vf = [torch.randn(1, 8, H // 2 // s, W // 2 // s, max(D // 2 // s, 1), generator=g) for s in (1, 2, 4)]
        gt = torch.randint(0, 19, (1, H, W, D), generator=g)
        gt[torch.rand(gt.shape, generator=g) < 0.55] = 18               # mostly free space
        gt[torch.rand(gt.shape, generator=g) < 0.10] = 255              # invisible voxels
        gt[gt == 7] = 18                                                 # a class absent from the targets
        with torch.no_grad():
            if tag == 'focal':
                logits = h(vf)['output_voxels'][0]
            losses = h.loss(output_voxels=[logits.clone()], target_voxels=gt)      # 'ce': same logits, CE_ssc_loss
        if tag == 'focal':
            for i, v in enumerate(vf):
                out[f'head.feat{i}'] = v.numpy()
            out['head.gt'] = gt.numpy().astype(np.uint8)
            out['head.logits_s5'] = logits[:, :, ::5, ::5].contiguous().numpy()      # every 5th BEV cell
            out['head.logits_sum'] = np.array(float(logits.double().sum()))
            for k, v in h.state_dict().items():
                out['w.head.' + k] = v.numpy()
            out['head.class_weights'] = h.class_weights.numpy()
        for k, v in losses.items():
            out[f'head.{tag}.{k}'] = np.array(float(v))
        print(tag, {k: float(v) for k, v in losses.items()})
    # gt at twice the head resolution: the majority-vote resize (:208-218)
    hsmall = head.OccHead(in_channels=[8], out_channel=19, num_level=1, soft_weights=False, use_focal_loss=False,
                          norm_cfg=dict(type='BN3d', requires_grad=True), final_occ_size=[8, 8, 4], empty_idx=18,
                          use_deblock=False)
    g = torch.Generator().manual_seed(8)
    gt2 = torch.randint(0, 19, (1, 8, 8, 4), generator=g)
    gt2[torch.rand(gt2.shape, generator=g) < 0.5] = 18
    lg = torch.randn(1, 19, 4, 4, 2, generator=g)
    with torch.no_grad():
        l2 = hsmall.loss(output_voxels=[lg.clone()], target_voxels=gt2.clone())
    out['resize.gt'], out['resize.logits'] = gt2.numpy(), lg.numpy()
    for k, v in l2.items():
        out[f'resize.{k}'] = np.array(float(v))

    # ---- image neck (shipped structure: 2 inputs, out_ids [0], one output)
    cf = fpn.CustomFPN(in_channels=[12, 24], out_channels=8, num_outs=1, start_level=0, out_ids=[0])
    _randomise(cf, 7)
    cf.eval()
    c4, c5 = torch.randn(2, 12, 6, 10), torch.randn(2, 24, 3, 5)
    with torch.no_grad():
        y = cf([c4, c5])
    out['fpn.c4'], out['fpn.c5'], out['fpn.out'] = c4.numpy(), c5.numpy(), y.numpy()
    for k, v in cf.state_dict().items():
        out['w.fpn.' + k] = v.numpy()

    # ---- parameter names / shapes of the in-tree blocks at the SHIPPED config (what a reference checkpoint contains)
    import json
    model = json.load(open(os.path.join(MG.OUT, 'fbocc_config_path_blocks.json')))['fbocc-r50-cbgs_depth_16f_16x4_20e.py']['model']
    strip = lambda d: {k: v for k, v in d.items() if k != 'type'}  # noqa: E731
    blocks = {'img_neck': fpn.CustomFPN(**strip(model['img_neck'])),
              'img_bev_encoder_backbone': r3d.CustomResNet3D(**strip(model['img_bev_encoder_backbone'])),
              'img_bev_encoder_neck': fpn3d.FPN3D(**strip(model['img_bev_encoder_neck'])),
              'occupancy_head': head.OccHead(**strip(model['occupancy_head']))}
    keys = {name: {k: list(v.shape) for k, v in m.state_dict().items()} for name, m in blocks.items()}
    json.dump(keys, open(os.path.join(MG.OUT, 'fbocc_reference_state_keys.json'), 'w'), indent=0, sort_keys=True)
    print('state keys:', {k: len(v) for k, v in keys.items()})

    path = os.path.join(MG.OUT, 'occ_encoder_head_small.npz')
    np.savez_compressed(path, **{k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim > 0 else v) for k, v in out.items()})
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
