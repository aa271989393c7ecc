"""Occupancy head of FB-OCC: `OccHead`.

Reference: mmdet3d/models/fbbev/heads/occupancy_head.py:24-266.  Inputs are the FPN3D levels (B, 256, Y_l, X_l, Z_l);
level 0 is up-sampled x2 by a transposed convolution (`deblock`, :129-141), every level goes through a 3x3x3 conv to
C/2 channels (:82-90), all four maps are brought to the finest resolution trilinearly and blended with per-voxel
softmax weights (:159-170), and a 1x1x1 MLP predicts `out_channel` class logits (:93-99).  Parameter names equal the
reference's (occ_convs.<i>.{0,1}, occ_pred_conv.{0,1,3}, voxel_soft_weights.{0,1,3}, deblock.{0,1}).

The convolutions stay on the vendor library; the losses are the sync-free restatements of occ_loss.py.
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import occ_loss as L
from .bev_encoder import build_norm, upsample_trilinear
from .mfma_conv3d import MConv3d, MConvTranspose3d


def _conv(conv_cfg, cin, cout, k, stride=1, padding=0):
    cfg = dict(conv_cfg)
    typ = cfg.pop('type')
    if typ == 'Conv3d':
        return MConv3d(cin, cout, k, stride=stride, padding=padding, **cfg)
    if typ == 'deconv3d':
        return MConvTranspose3d(cin, cout, k, stride=stride, padding=padding, **cfg)
    raise KeyError(typ)


class OccHead(nn.Module):
    def __init__(self, in_channels, out_channel, num_level=1, soft_weights=False, loss_weight_cfg=None,
                 conv_cfg=dict(type='Conv3d', bias=False), norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), final_occ_size=(256, 256, 20), empty_idx=0,
                 balance_cls_weight=True, train_cfg=None, test_cfg=None, with_cp=False, use_focal_loss=False,
                 use_dice_loss=False, use_deblock=True, compute_dtype=torch.float32):
        super().__init__()
        if use_dice_loss:
            raise NotImplementedError('DiceLoss (:114-116) is mmseg-external and unused by the fb_occ configs')
        in_channels = list(in_channels) if isinstance(in_channels, (list, tuple)) else [in_channels]
        self.in_channels, self.out_channel, self.num_level = in_channels, out_channel, num_level
        self.with_cp, self.use_deblock, self.use_focal_loss = with_cp, use_deblock, use_focal_loss
        self.soft_weights, self.empty_idx, self.compute_dtype = soft_weights, empty_idx, compute_dtype
        self.final_occ_size = list(final_occ_size)
        if use_focal_loss:
            self.focal_loss = L.CustomFocalLoss(bev_hw=tuple(self.final_occ_size[:2]))
        cfg = loss_weight_cfg or {}
        self.loss_voxel_ce_weight = cfg.get('loss_voxel_ce_weight', 1.0)
        self.loss_voxel_sem_scal_weight = cfg.get('loss_voxel_sem_scal_weight', 1.0)
        self.loss_voxel_geo_scal_weight = cfg.get('loss_voxel_geo_scal_weight', 1.0)
        self.loss_voxel_lovasz_weight = cfg.get('loss_voxel_lovasz_weight', 1.0)

        norm = lambda c: build_norm(norm_cfg, c)[1]  # noqa: E731
        self.occ_convs = nn.ModuleList()
        for i in range(num_level):
            mid = in_channels[i] // 2
            self.occ_convs.append(nn.Sequential(_conv(conv_cfg, in_channels[i], mid, 3, padding=1), norm(mid),
                                                nn.ReLU(inplace=True)))
        self.occ_pred_conv = nn.Sequential(_conv(conv_cfg, mid, mid // 2, 1), norm(mid // 2), nn.ReLU(inplace=True),
                                           _conv(conv_cfg, mid // 2, out_channel, 1))
        self.num_point_sampling_feat = num_level + (1 if use_deblock else 0)
        if soft_weights:
            self.voxel_soft_weights = nn.Sequential(_conv(conv_cfg, mid, mid // 2, 1), norm(mid // 2),
                                                    nn.ReLU(inplace=True),
                                                    _conv(conv_cfg, mid // 2, self.num_point_sampling_feat, 1))
        self.register_buffer('class_weights', L.class_weights(out_channel, balance_cls_weight).float(), persistent=False)
        if use_deblock:
            self.deblock = nn.Sequential(_conv(dict(type='deconv3d', bias=False), in_channels[0], in_channels[0] // 2, 2,
                                               stride=2), norm(in_channels[0] // 2), nn.ReLU(inplace=True))

    def _run(self, mod, x):
        return checkpoint(mod, x, use_reentrant=False) if (self.with_cp and x.requires_grad) else mod(x)

    def forward_coarse_voxel(self, voxel_feats):
        """:143-181."""
        output_occs = []
        if self.use_deblock:
            output_occs.append(self._run(self.deblock, voxel_feats[0]))
        for feats, conv in zip(voxel_feats, self.occ_convs):
            output_occs.append(self._run(conv, feats))
        if self.soft_weights:
            w = torch.softmax(self.voxel_soft_weights(output_occs[0]), dim=1)
        else:
            w = output_occs[0].new_ones(output_occs[0].shape[0], self.num_point_sampling_feat, 1, 1, 1) \
                / self.num_point_sampling_feat
        size = output_occs[0].shape[2:]
        out = None
        for feats, wk in zip(output_occs, torch.unbind(w, dim=1)):
            if tuple(feats.shape[2:]) != tuple(size):          # trilinear resize to the same size is the identity
                feats = upsample_trilinear(feats, size)
            # out += feats * weights (:169) as one multiply-add pass per level instead of a multiply and an add pass over
            # the full-resolution 128-channel maps
            out = feats * wk.unsqueeze(1) if out is None else torch.addcmul(out, feats, wk.unsqueeze(1))
        return {'out_voxel_feats': [out], 'occ': [self._run(self.occ_pred_conv, out)]}

    def forward(self, voxel_feats, img_feats=None, pts_feats=None, transform=None, **kwargs):
        assert isinstance(voxel_feats, (list, tuple)) and len(voxel_feats) == self.num_level
        if self.compute_dtype != torch.float32 and voxel_feats[0].is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                output = self.forward_coarse_voxel(voxel_feats)
            output['occ'] = [o.float() for o in output['occ']]
        else:
            output = self.forward_coarse_voxel([v.float() for v in voxel_feats])
        return {'output_voxels': output['occ'], 'output_voxels_fine': None, 'output_coords_fine': None}

    def forward_train(self, voxel_feats, img_feats=None, pts_feats=None, transform=None, gt_occupancy=None,
                      gt_occupancy_flow=None, **kwargs):
        res = self.forward(voxel_feats, img_feats=img_feats, pts_feats=pts_feats, transform=transform, **kwargs)
        return self.loss(target_voxels=gt_occupancy, output_voxels=res['output_voxels'])

    def _resize_gt(self, target_voxels, B, H, W, D, ratio):
        """:208-218 -- majority vote inside each ratio^3 cell (all-empty cells stay empty; free voxels (label 0) never
        win a vote against an occupied label; a cell with only free voxels becomes 255)."""
        t = target_voxels.reshape(B, H, ratio, W, ratio, D, ratio).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, H, W, D, ratio ** 3)
        empty_mask = t.sum(-1) == self.empty_idx
        t = t.to(torch.int64)
        occ_space = t[~empty_mask]
        zeros = occ_space == 0
        occ_space[zeros] = -torch.arange(int(zeros.sum()), device=t.device) - 1
        t[~empty_mask] = occ_space
        t = torch.mode(t, dim=-1)[0]
        t[t < 0] = 255
        return t.long()

    def loss_voxel(self, output_voxels, target_voxels, tag):
        """:200-246."""
        B, C, H, W, D = output_voxels.shape
        ratio = target_voxels.shape[2] // H
        if ratio != 1:
            target_voxels = self._resize_gt(target_voxels, B, H, W, D, ratio)
        output_voxels = torch.nan_to_num(output_voxels.float(), nan=0.0, posinf=0.0, neginf=0.0)   # :222-223
        target_voxels = target_voxels.long()
        cw = self.class_weights.to(output_voxels)
        loss = {}
        if self.use_focal_loss:
            ce = self.focal_loss(output_voxels, target_voxels, cw, ignore_index=255)
        else:
            ce = L.CE_ssc_loss(output_voxels, target_voxels, cw, ignore_index=255)
        loss[f'loss_voxel_ce_{tag}'] = self.loss_voxel_ce_weight * ce
        loss[f'loss_voxel_sem_scal_{tag}'] = self.loss_voxel_sem_scal_weight * L.sem_scal_loss(
            output_voxels, target_voxels, ignore_index=255)
        loss[f'loss_voxel_geo_scal_{tag}'] = self.loss_voxel_geo_scal_weight * L.geo_scal_loss(
            output_voxels, target_voxels, ignore_index=255, non_empty_idx=self.empty_idx)
        loss[f'loss_voxel_lovasz_{tag}'] = self.loss_voxel_lovasz_weight * L.lovasz_softmax(
            torch.softmax(output_voxels, dim=1), target_voxels, ignore=255)
        return loss

    def loss(self, output_voxels=None, output_coords_fine=None, output_voxels_fine=None, target_voxels=None,
             visible_mask=None, **kwargs):
        out = {}
        for i, ov in enumerate(output_voxels):
            out.update(self.loss_voxel(ov, target_voxels, tag=f'c_{i}'))
        return out
