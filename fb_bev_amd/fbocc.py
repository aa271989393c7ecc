"""The FB-OCC detector assembled around the MI355X view-transformation path (scopes S4 / S5).

Reference: mmdet3d/models/fbbev/detectors/fbocc.py
    :47-131    __init__: sub-module blocks of the `model` config dict (img_backbone / img_neck come from the
               CenterPoint -> MVXTwoStageDetector bases, external mmdet3d)
    :135-149   image_encoder            :151-162  bev_encoder
    :322-376   extract_img_bev_feat     (image encoder -> depth net -> forward projection -> backward projection ->
                                         re-add -> history fusion -> voxel encoder)
    :400-461   forward_train            (occupancy losses + depth loss)
    :463-505   forward_test             :516-598  simple_test (softmax, CVPR-2023 axis convention, argmax)

`FBOCC(**cfg['model'])` takes the reference's config block unchanged (occupancy_configs/fb_occ/*.py) and keeps the
reference's parameter names (img_backbone.*, img_neck.*, depth_net.*, backward_projection.*,
history_keyframe_{time,cat}_conv.*, img_bev_encoder_{backbone,neck}.*, occupancy_head.*), so a reference checkpoint's
state dict loads key for key.

What runs where: the view transformation, history alignment and the attention kernels are this repo's HIP kernels
(fb_view_transform.py, history_fusion.py); the dense convolution stacks (image / voxel encoders, depth net, head) are
MFMA-bound and run on MIOpen with channels-last activations and, optionally, bf16 (`execution` knobs).  The occupancy
losses are the sync-free restatements of occ_loss.py.  LiDAR branches, `frpn` and `pts_bbox_head` of the reference
class are not used by any fb_occ config and are rejected.
"""
import torch
import torch.nn as nn

from .bev_encoder import CustomResNet3D, FPN3D
from .depth_net import CM_DepthNet
from .fb_view_transform import FBViewTransform
from .history_fusion import TemporalHistoryFusion
from .img_encoder import CustomFPN, ResNet
from .occ_head import OccHead

_TYPES = {'ResNet': ResNet, 'CustomFPN': CustomFPN, 'CM_DepthNet': CM_DepthNet, 'CustomResNet3D': CustomResNet3D,
          'FPN3D': FPN3D, 'OccHead': OccHead}


def _build(cfg, **extra):
    if cfg is None:
        return None
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ not in _TYPES:
        raise KeyError(f'{typ!r} is not a block of the FB-OCC occupancy configs')
    cfg.update(extra)
    return _TYPES[typ](**cfg)


def _dtype(v):
    return {None: torch.float32, 'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16,
            'f16': torch.float16, 'fp16': torch.float16}.get(v, v)


class FBOCC(nn.Module):
    def __init__(self, forward_projection=None, img_bev_encoder_backbone=None, img_bev_encoder_neck=None,
                 backward_projection=None, frpn=None, depth_net=None, occupancy_head=None, use_depth_supervision=False,
                 readd=False, fix_void=False, occupancy_save_path=None, do_history=True, interpolation_mode='bilinear',
                 history_cat_num=16, history_cat_conv_out_channels=None, single_bev_num_channels=80, img_backbone=None,
                 img_neck=None, pts_bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None,
                 execution=None, **kwargs):
        """execution (not part of the reference config): dict(img_dtype=, depth_dtype=, voxel_dtype=, head_dtype=) with
        values 'f32' | 'bf16' -- compute dtype of the four convolution stacks (default fp32 everywhere; the shipped
        config trains them under mmcv's fp16 hook, cfg :394) -- and with_cp=True|False to override the blocks'
        activation checkpointing (the configs turn it on to fit 16-32 GB parts; 288 GB of HBM does not need it);
        history_dtype='f16' | 'bf16' stores the inference history ring in 16 bits (BASELINE configs[4]);
        history_compute='bf16' runs the two fused history convolutions on the bf16 MFMA (fp32 accumulate) at inference,
        'bf16x3' the same with split operands (three MFMAs per product: 6.5e-6 of the output peak against the fp32 convolutions;
        16-bit voxel-major ring), 'f32' the reference's arithmetic; default 'auto' = 'bf16x3' for a 16-bit ring, 'f32' otherwise;
        history_ring='voxel_major' keeps the inference ring as (B, T, N, C) voxel rows (16-byte taps, same element bits);
        da_value_dtype='bf16' | 'f16' keeps the cross-attention's camera tokens in 16 bits at inference (fp32 accumulate);
        mfma_conv3d / mfma_conv3d_train=True route the voxel encoder + head through fbbev_conv3d_* (mfma_conv3d.py)."""
        super().__init__()
        if frpn is not None or pts_bbox_head is not None:
            raise NotImplementedError('frpn / pts_bbox_head are None in every fb_occ config (fbocc.py:86-88 "not used in FB-OCC")')
        if forward_projection is None:
            raise ValueError('FBOCC needs a forward_projection block')
        ex = dict(execution or {})
        cp = {} if ex.get('with_cp') is None else {'with_cp': bool(ex['with_cp'])}
        self.fix_void, self.readd, self.use_depth_supervision = fix_void, readd, use_depth_supervision
        self.occupancy_save_path = occupancy_save_path
        self.img_backbone = _build(img_backbone, **cp, compute_dtype=_dtype(ex.get('img_dtype')))
        self.img_neck = _build(img_neck, **cp, compute_dtype=_dtype(ex.get('img_dtype')))
        self.depth_net = _build(depth_net, **cp, compute_dtype=_dtype(ex.get('depth_dtype')))
        fvt = FBViewTransform(forward_projection, backward_projection, readd=readd)
        self.forward_projection = fvt.forward_projection
        self.backward_projection = fvt.backward_projection
        if ex.get('da_value_dtype') and self.backward_projection is not None:   # 'bf16' | 'f16': 16-bit camera tokens (inference)
            from .backward_projection import DA_SpatialCrossAttention
            for mod in self.backward_projection.modules():
                if isinstance(mod, DA_SpatialCrossAttention):
                    mod.value_dtype = _dtype(ex['da_value_dtype'])
        fp = fvt.forward_projection
        hist = TemporalHistoryFusion(fp.dx.tolist(), fp.bx.tolist(), single_bev_num_channels=single_bev_num_channels,
                                     history_cat_num=history_cat_num,
                                     history_cat_conv_out_channels=history_cat_conv_out_channels, do_history=do_history,
                                     interpolation_mode=interpolation_mode, history_dtype=_dtype(ex.get('history_dtype')),
                                     history_compute=_dtype(ex['history_compute']) if ex.get('history_compute') else 'auto',
                                     ring_layout=ex.get('history_ring', 'voxel_major'))
        # the reference registers the two fusion convolutions on the detector itself (fbocc.py:111-127): same names here
        self.history_keyframe_time_conv = hist.history_keyframe_time_conv
        self.history_keyframe_cat_conv = hist.history_keyframe_cat_conv
        self._path = [fvt, hist]                  # plain list: holders stay out of the parameter tree (names above)
        self.img_bev_encoder_backbone = _build(img_bev_encoder_backbone, **cp, compute_dtype=_dtype(ex.get('voxel_dtype')))
        self.img_bev_encoder_neck = _build(img_bev_encoder_neck, **cp, compute_dtype=_dtype(ex.get('voxel_dtype')))
        self.occupancy_head = _build(occupancy_head, **cp, compute_dtype=_dtype(ex.get('head_dtype')))
        # eval-mode image encoder / voxel encoder / head on the hand-written implicit-GEMM kernels (mfma_conv3d.py).
        # GPU-validated in round 2 (tests/test_gpu_conv3d.py) and measured against the vendor route on the shipped
        # config (profiles/r02_time_full.jsonl: S4 22.2 ms fp32-MFMA vs 41.1 ms vendor bf16), so exact-fp32 MFMA is
        # the default; 'bf16' / 'bf16_tiled' are the faster reduced-precision settings, False = vendor library.
        self.mfma_conv3d = ex.get('mfma_conv3d', True)           # False | True (fp32 MFMA) | 'bf16' | 'bf16_tiled' (bf16 MFMA where Cin % 32 == 0)
        self._runners = None
        self._runner_tensors = self._runner_key_built = None
        if ex.get('mfma_conv3d_train'):           # opt-in: the autograd route (forward + dgrad + wgrad kernels)
            from .mfma_conv3d import enable_training_route
            for blk in (self.img_bev_encoder_backbone, self.img_bev_encoder_neck, self.occupancy_head,
                        self.history_keyframe_time_conv, self.history_keyframe_cat_conv):
                if blk is not None:
                    enable_training_route(blk, True)

    # ------------------------------------------------------------------ plumbing
    @property
    def view_transform(self):
        return self._path[0]

    @property
    def history(self):
        return self._path[1]

    @property
    def do_history(self):
        return self.history.do_history

    @do_history.setter
    def do_history(self, v):
        self.history.do_history = bool(v)

    def train(self, mode=True):
        super().train(mode)
        for m in self._path:
            m.training = mode
        self._runners = None                      # folded weights are rebuilt from the current parameters on next use
        return self

    def _apply(self, fn, *args, **kwargs):
        self._runners = None                      # .to() / .cuda() / .half(): the snapshots point at the old storage
        return super()._apply(fn, *args, **kwargs)

    def invalidate_folded_weights(self):
        """Drop the BN-folded weight snapshots of the MFMA convolution runners.  They are keyed on (storage pointer,
        in-place version) of every parameter / buffer they fold, so load_state_dict(), optimizer steps, .to() and
        `p.copy_()` are noticed by themselves; a write through `p.data` (some EMA implementations) bumps no version
        counter -- call this after such a swap."""
        self._runners = None

    def _runner_key(self):
        ts = self._runner_tensors
        return tuple(t.data_ptr() for t in ts), sum(t._version for t in ts)

    def _mfma_stacks(self):
        # the runners snapshot BN-folded weights: rebuild them when any folded tensor was replaced or written in place
        # (eval -> forward -> load_state_dict / EMA swap -> forward must not run on the old weights)
        if self._runners is not None and self._runner_tensors is not None and self._runner_key() != self._runner_key_built:
            self._runners = None
        if self._runners is None:
            from . import mfma_conv3d as M
            prec = self.mfma_conv3d if self.mfma_conv3d in ('bf16', 'bf16_tiled') else 'f32'
            img = None
            if self.img_backbone is not None and self.img_neck is not None:
                try:
                    img = (M.ResNetRunner(self.img_backbone, prec), M.CustomFPNRunner(self.img_neck, prec))
                except (ValueError, NotImplementedError):      # channel counts / layers outside the kernel: vendor route
                    img = None
            try:
                self._runners = (M.ResNet3DRunner(self.img_bev_encoder_backbone, prec), M.FPN3DRunner(self.img_bev_encoder_neck, prec),
                                 M.OccHeadRunner(self.occupancy_head, prec), img)
            except (ValueError, NotImplementedError, AttributeError, TypeError) as e:
                # a block outside what the kernels cover (channel counts, norm type, missing block): vendor route
                import warnings
                warnings.warn(f'FBOCC: convolution stacks stay on the vendor library ({e})')
                self.mfma_conv3d = False
                self._runners = (None, None, None, None)
            blocks = [self.img_bev_encoder_backbone, self.img_bev_encoder_neck, self.occupancy_head]
            if img is not None:
                blocks += [self.img_backbone, self.img_neck]
            self._runner_tensors = None
            if any(r is not None for r in self._runners):
                self._runner_tensors = [t for b in blocks if b is not None for t in list(b.parameters()) + list(b.buffers())]
                self._runner_key_built = self._runner_key()
        return self._runners

    def _use_mfma(self, x):
        if not (bool(self.mfma_conv3d) and x.is_cuda and not self.training and not torch.is_grad_enabled()):
            return False
        self._mfma_stacks()                       # may turn the route off for blocks outside the kernels
        return bool(self.mfma_conv3d)

    def reset_history(self):
        self.history.reset()

    # ------------------------------------------------------------------ fbocc.py:135-162
    def image_encoder(self, img):
        B, N, C, H, W = img.shape
        if self._use_mfma(img) and self._mfma_stacks()[3] is not None:
            backbone, neck = self._mfma_stacks()[3]
            x = neck(backbone(img.view(B * N, C, H, W)))
            return x.reshape(B, N, *x.shape[1:])
        x = self.img_backbone(img.view(B * N, C, H, W))
        if self.img_neck is not None:
            x = self.img_neck(x)
            if isinstance(x, (list, tuple)):
                x = x[0]
        return x.view(B, N, *x.shape[1:])

    def bev_encoder(self, x):
        if self.img_bev_encoder_backbone is not None:
            x = self.img_bev_encoder_backbone(x)
        if self.img_bev_encoder_neck is not None:
            x = self.img_bev_encoder_neck(x)
        return list(x) if isinstance(x, (list, tuple)) else [x]

    # ------------------------------------------------------------------ fbocc.py:322-376
    def extract_img_bev_feat(self, img, img_metas, **kwargs):
        ret = {}
        context = self.image_encoder(img[0])
        cam_params = list(img[1:7])
        depth = None
        if self.depth_net is not None:
            mlp_input = self.depth_net.get_mlp_input(*cam_params)
            context, depth = self.depth_net(context, mlp_input)
            ret['depth'], ret['context'] = depth, context
        bev_feat = self.view_transform(cam_params, context.float(), depth.float(), img_metas=img_metas)   # :344-368
        ret['cam_params'] = cam_params
        bev_feat = self.history.fuse_history(bev_feat, img_metas, img[6])                                # :371
        if self._use_mfma(bev_feat):
            from .mfma_conv3d import to_ndhwc
            backbone, neck = self._mfma_stacks()[:2]
            ret['img_bev_feat_ndhwc'] = neck(backbone(to_ndhwc(bev_feat)))       # NDHWC maps, consumed by the head runner
            return ret
        ret['img_bev_feat'] = self.bev_encoder(bev_feat)
        return ret

    def extract_feat(self, points, img, img_metas, **kwargs):
        return self.extract_img_bev_feat(img, img_metas, **kwargs) if img is not None else {}

    # ------------------------------------------------------------------ fbocc.py:400-461
    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img_inputs=None, proposals=None, gt_bboxes_ignore=None, gt_occupancy_flow=None,
                      **kwargs):
        results = self.extract_feat(points, img=img_inputs, img_metas=img_metas, **kwargs)
        losses = {}
        if self.occupancy_head is not None:
            losses.update(self.occupancy_head.forward_train(results['img_bev_feat'], results=results,
                                                            gt_occupancy=kwargs['gt_occupancy'],
                                                            gt_occupancy_flow=gt_occupancy_flow))
        if self.use_depth_supervision and self.depth_net is not None:
            losses.update(self.depth_net.get_depth_loss(kwargs['gt_depth'], results['depth']))
        return losses

    @staticmethod
    def parse_losses(losses):
        """mmdet BaseDetector._parse_losses (external): total = sum of every entry whose key contains 'loss'."""
        return sum(v.mean() for k, v in losses.items() if 'loss' in k)

    # ------------------------------------------------------------------ fbocc.py:463-598
    def forward_test(self, points=None, img_metas=None, img_inputs=None, **kwargs):
        self.do_history = True                                                          # :485
        for var, name in ((img_inputs, 'img_inputs'), (img_metas, 'img_metas')):
            if not isinstance(var, list):
                raise TypeError('{} must be a list, but got {}'.format(name, type(var)))
        if len(img_inputs) != len(img_metas):
            raise ValueError('num of augmentations ({}) != num of image meta ({})'.format(len(img_inputs), len(img_metas)))
        if len(img_inputs) != 1:
            raise NotImplementedError('test-time augmentation: the reference asserts False in aug_test (fbocc.py:513)')
        return self.simple_test(None if points is None else points[0], img_metas[0], img_inputs[0], **kwargs)

    def predict_occupancy(self, img_inputs, img_metas, return_raw_occ=False, **kwargs):
        """simple_test's device-side part for a whole batch: -> (B, X', Y', Z) class ids (or (B, X', Y', Z, classes)
        probabilities) in the CVPR-2023 axis convention of :547-552, still on the GPU."""
        results = self.extract_feat(None, img=img_inputs, img_metas=img_metas, **kwargs)
        if 'img_bev_feat_ndhwc' in results:
            occ = self._mfma_stacks()[2](results['img_bev_feat_ndhwc'])                           # (B,cls,H,W,D) view
        else:
            occ = self.occupancy_head(results['img_bev_feat'], results=results)['output_voxels'][0]   # (B,cls,H,W,D)
        if self.fix_void:
            occ = occ[:, 1:]                                                            # :542-543
        occ = occ.softmax(1)
        # :547-552 on sample 0 of permute(0,2,3,4,1): permute(3,2,0,1) -> flip(dim H) -> rot90(-1,[H,W]) -> permute(2,3,1,0).
        # The class axis only rides along, so argmax is taken first and the spatial shuffle runs on one id per voxel.
        x = occ if return_raw_occ else occ.argmax(1, keepdim=True)                      # (B,c,H,W,D)
        x = x.permute(0, 1, 4, 2, 3)                                                    # (B,c,D,H,W)
        x = torch.rot90(torch.flip(x, [3]), -1, [3, 4])
        x = x.permute(0, 3, 4, 2, 1)                                                    # (B,H',W',D,c)
        return x if return_raw_occ else x[..., 0]

    def simple_test(self, points, img_metas, img=None, rescale=False, visible_mask=(None,), return_raw_occ=False, **kwargs):
        assert len(img_metas) == 1                                                      # :591
        pred = self.predict_occupancy(img, img_metas, return_raw_occ=return_raw_occ, **kwargs)[0]
        return [dict(pts_bbox=None, iou=None, pred_occupancy=pred.cpu().numpy(), index=img_metas[0].get('index'))]

    def forward(self, return_loss=True, **kwargs):
        """mmdet BaseDetector.forward dispatch (external)."""
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)
