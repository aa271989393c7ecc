"""Voxel encoder behind the view transformation: `CustomResNet3D` + `FPN3D`.

Reference: mmdet3d/models/fbbev/modules/resnet3d.py:46-274 (backbone; BasicBlock :46-102, trunk
:143-274) and mmdet3d/models/fbbev/modules/fpn3d.py:14-110 (neck), called from FBOCC.bev_encoder (fbocc.py:151-162) on
the (B, C, Y, X, Z) volume the view transformation produces.  The dense (non-spconv) branch only: the shipped configs
never set use_spase_3dtensor.  Parameter names equal the reference's, so a detector checkpoint loads by prefix:
    input_proj.{0,1}.*  layers.<i>.<j>.{conv1,bn1,conv2,bn2,downsample.{0,1}}.*      (backbone)
    lateral_convs.<i>.0.{conv,bn|gn}.*  fpn_convs.<i>.0.{conv,bn|gn}.*               (neck; mmcv ConvModule naming)

These are dense 3-D convolutions: MFMA-bound work that stays on the vendor library (MIOpen), as SURVEY 8a row 1 / 8f-3
prescribe.  What this file adds over the reference is the execution setup for MI355X: activations kept in
channels_last_3d (NDHWC, the layout MIOpen's implicit-GEMM kernels read without transposes) and an optional bf16
compute dtype (`compute_dtype`, default fp32 = the reference's @force_fp32).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .mfma_conv3d import MConv3d


def _linear_resize_matrix(n_in, n_out, device):
    """(n_out, n_in) matrix of ATen's 1-D linear resize, align_corners=False: src = (dst + 0.5) * n_in / n_out - 0.5,
    clamped at 0; taps floor(src) and the next index (clamped at n_in - 1) with weights 1 - frac, frac."""
    dst = torch.arange(n_out, dtype=torch.float32, device=device)
    src = ((dst + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp_(min=0)
    i0 = src.floor().clamp_(max=n_in - 1)
    lam = src - i0
    i0 = i0.long()
    i1 = (i0 + 1).clamp_(max=n_in - 1)
    w = torch.zeros(n_out, n_in, dtype=torch.float32, device=device)
    rows = torch.arange(n_out, device=device)
    w.index_put_((rows, i0), 1.0 - lam, accumulate=True)
    w.index_put_((rows, i1), lam, accumulate=True)
    return w


def upsample_trilinear(x, size):
    """F.interpolate(x, size, mode='trilinear', align_corners=False) (occupancy_head.py:163-166, fpn3d.py:92-96).  Under autograd on the
    GPU the resize is applied as its three 1-D factors (trilinear interpolation is separable), each a small dense
    matrix applied FROM THE LEFT to the tensor viewed as (outer, n_in, inner): the same linear map (to fp32 summation order),
    whose backward is the transposed matrix applied the same way instead of ATen's atomic scatter
    (upsample_trilinear3d_backward: 63 % of the training step at 200x200x16 before this,
    profiles/r02_rocprofv3_train_step_before_separable_upsample.csv).  Round 3: no transposes -- the round-2 form moved the
    resized axis to the end for `x @ W^T` and back, and every hop copied the (4, 128, 200, 200, 16) maps of the head: 49 ms
    of the 303 ms training step (profiles/r03_train_step_copy_sites.json).  The axes are resized in the MEMORY order of x
    (channels_last_3d stays channels_last_3d: the layout the convolution kernels produce and consume)."""
    if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return F.interpolate(x, size=list(size), mode='trilinear', align_corners=False)
    cl = x.dim() == 5 and not x.is_contiguous() and x.permute(0, 2, 3, 4, 1).is_contiguous()
    base = (x.permute(0, 2, 3, 4, 1) if cl else x.contiguous()).float()       # contiguous: (B,d0,d1,d2,C) or (B,C,d0,d1,d2)
    first = 1 if cl else 2                                                     # position of the first spatial axis in `base`
    for k, n_out in enumerate(size):                                           # smallest tensors first
        axis = first + k
        n_in = base.shape[axis]
        if n_in != n_out:
            w = _linear_resize_matrix(n_in, n_out, x.device)
            shape = list(base.shape)
            outer = 1
            for d in shape[:axis]:
                outer *= d
            y = torch.matmul(w, base.reshape(outer, n_in, -1))                 # (outer, n_out, inner): contiguous, no transposes
            shape[axis] = n_out
            base = y.view(shape)
    out = base.permute(0, 4, 1, 2, 3) if cl else base
    return out.to(x.dtype)


def batch_norm_channels_last_3d(bn_forward, x):
    """Run a batch-norm forward (a callable taking a 2-D (M, C) tensor) on a 5-D tensor that lives in channels_last_3d
    memory -- what MConv3d / fbbev_conv3d_* produce -- WITHOUT the NCDHW round trip: the (B, D, H, W, C) rows are a
    contiguous (M, C) matrix, batch norm over dim 0 of it is the same per-channel statistics, and the result goes back as the
    same permuted view.  The vendor batch norm takes NCDHW only: every conv -> norm -> conv hop of the 3-D stacks cost two
    full-tensor copies forward and two backward (49 ms of the 303 ms training step, 221 launches,
    profiles/r03_train_step_kernels_bf16.json)."""
    B, C, D, H, W = x.shape
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, C)                 # a view: no copy
    with torch.backends.cudnn.flags(enabled=False):                # ATen's channels-last kernels (the vendor path wants NC(D)HW)
        y = bn_forward(rows)
    return y.view(B, D, H, W, C).permute(0, 4, 1, 2, 3)


def _is_channels_last_3d(x):
    return x.dim() == 5 and x.is_cuda and not x.is_contiguous() and x.permute(0, 2, 3, 4, 1).is_contiguous()


class BatchNorm3d(nn.BatchNorm3d):
    """nn.BatchNorm3d (same parameters, buffers, state-dict names and arithmetic) that keeps a channels_last_3d activation in
    its layout instead of converting it to NCDHW and back."""

    def _check_input_dim(self, x):
        if x.dim() not in (2, 5):
            raise ValueError(f'expected 5D input (got {x.dim()}D input)')

    def forward(self, x):
        if _is_channels_last_3d(x):
            return batch_norm_channels_last_3d(super().forward, x)
        return super().forward(x)


def build_norm(norm_cfg, channels, dims=3):
    """mmcv.cnn.build_norm_layer (external) for the types the FB-OCC configs use -> (state-dict abbreviation, layer).
    SyncBN is built as plain BatchNorm (identical parameters / state names / eval arithmetic) and MARKED
    (`_fbbev_sync_bn`): `shard.convert_sync_batchnorm(model)` -- called by `shard.prepare_ddp` for every multi-rank
    training job -- swaps the marked layers for the cross-rank implementation, as the reference's SyncBN does."""
    cfg = dict(norm_cfg or dict(type='BN'))
    typ = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    if typ in ('BN', 'BN1d', 'BN2d', 'BN3d', 'SyncBN'):
        cls = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: BatchNorm3d}[dims]
        if typ == 'BN1d':
            cls = nn.BatchNorm1d
        elif typ == 'BN2d':
            cls = nn.BatchNorm2d
        elif typ == 'BN3d':
            cls = BatchNorm3d
        cfg.setdefault('eps', 1e-5)
        layer, abbr = cls(channels, **cfg), 'bn'
        layer._fbbev_sync_bn = (typ == 'SyncBN')
    elif typ == 'GN':
        cfg.setdefault('eps', 1e-5)
        layer, abbr = nn.GroupNorm(num_channels=channels, **cfg), 'gn'
    else:
        raise KeyError(f'norm type {typ!r} is not used by the FB-OCC configs')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr, layer


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule (external) restricted to order conv -> norm -> act; children named conv / bn|gn / activate
    like mmcv's, bias='auto' (a bias only when there is no norm)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), bias='auto', inplace=True):
        super().__init__()
        ctype = (conv_cfg or dict(type='Conv2d'))['type']
        conv = {'Conv2d': nn.Conv2d, 'Conv3d': MConv3d, 'Conv': nn.Conv2d}[ctype]
        dims = 3 if ctype == 'Conv3d' else 2
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = conv(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')   # ConvModule.init_weights
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = build_norm(norm_cfg, out_channels, dims)
            self.add_module(self.norm_name, norm)
        self.activate = None
        if act_cfg is not None:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.norm_name is not None:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


def _conv3(cin, cout, stride=1):
    return MConv3d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=True)     # resnet3d.py:19-30 (BIAS = True)


def _conv1(cin, cout, stride=1):
    return MConv3d(cin, cout, kernel_size=1, stride=stride, bias=True)                # :33-43


class BasicBlock3D(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None, norm_cfg=None):
        super().__init__()
        self.conv1 = _conv3(in_planes, planes, stride)
        self.bn1 = build_norm(norm_cfg, planes)[1]
        self.conv2 = _conv3(planes, planes)
        self.bn2 = build_norm(norm_cfg, planes)[1]
        self.downsample = downsample

    def forward(self, x):                                                              # :78-102
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return F.relu(out + residual)


def _low_precision_ok(dtype, x):
    """bf16 for the voxel encoder is an INFERENCE setting: the backward pass of the bf16 3-D convolutions of this stack
    segfaults inside the vendor library on the ROCm 7.2 image (profiles/r01_full_model_bf16_voxel_backward_crash.log;
    the 2-D stacks and the occupancy head train in bf16), so under autograd the stack computes in fp32."""
    return dtype != torch.float32 and x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad)


class CustomResNet3D(nn.Module):
    """resnet3d.py:143-274.  `depth` picks the block type and the per-stage block counts (:159-172); only the first
    len(block_inplanes) stages are built (:198-200)."""

    def __init__(self, depth, block_inplanes=(64, 128, 256, 512), block_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 n_input_channels=3, shortcut_type='B', with_cp=False, norm_cfg=dict(type='BN3d', requires_grad=True),
                 use_spase_3dtensor=False, plane2voxel=None, widen_factor=1.0, channels_last=True,
                 compute_dtype=torch.float32):
        super().__init__()
        if use_spase_3dtensor:
            raise NotImplementedError('the spconv branch (resnet3d.py:56-66) is not used by any fb_occ config')
        counts = {10: [1, 1, 1, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}[depth]
        if depth not in (10, 18, 34):
            # the reference cannot build its Bottleneck variants either: _make_layer (:237-243) passes
            # use_spase_3dtensor to Bottleneck.__init__ (:110), which does not accept it -> TypeError
            raise NotImplementedError('CustomResNet3D depth 50/101 is not constructible in the reference')
        block = BasicBlock3D
        if shortcut_type != 'B':
            raise NotImplementedError("shortcut_type 'A' (:213-222) is not used by any fb_occ config")
        self.with_cp, self.plane2voxel, self.out_indices = with_cp, plane2voxel, tuple(out_indices)
        self.channels_last, self.compute_dtype = channels_last, compute_dtype
        planes = [int(p * widen_factor) for p in block_inplanes]
        self.in_planes = planes[0]
        self.input_proj = nn.Sequential(MConv3d(n_input_channels, self.in_planes, kernel_size=1, bias=False),
                                        build_norm(norm_cfg, self.in_planes)[1], nn.ReLU(inplace=True))
        self.layers = nn.ModuleList(
            self._make_layer(block, planes[i], counts[i], block_strides[i], norm_cfg) for i in range(len(planes)))
        for m in self.modules():                                                       # :202-211
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride, norm_cfg):
        downsample = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            downsample = nn.Sequential(_conv1(self.in_planes, planes * block.expansion, stride),
                                       build_norm(norm_cfg, planes * block.expansion)[1])
        layers = [block(self.in_planes, planes, stride, downsample, norm_cfg)]
        self.in_planes = planes * block.expansion
        layers += [block(self.in_planes, planes, norm_cfg=norm_cfg) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _stages(self, x):
        x = self.input_proj(x)
        res = []
        for i, layer in enumerate(self.layers):
            x = checkpoint(layer, x, use_reentrant=False) if (self.with_cp and x.requires_grad) else layer(x)
            if i in self.out_indices:
                res.append(x)
        return res

    def forward(self, x):
        if self.plane2voxel is not None:
            x = x.unsqueeze(-1).repeat(1, 1, 1, 1, self.plane2voxel)
        if self.channels_last and x.is_cuda:        # MIOpen layout; ATen's CPU NDHWC backward is not relied on
            x = x.contiguous(memory_format=torch.channels_last_3d)
        if _low_precision_ok(self.compute_dtype, x):
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._stages(x)
        return self._stages(x.float())


class FPN3D(nn.Module):
    """fpn3d.py:14-110: 1x1x1 lateral ConvModules, top-down trilinear (align_corners=False) merge, 3x3x3 output
    ConvModules, one output per input level."""

    def __init__(self, in_channels=(80, 160, 320, 640), out_channels=256,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), conv_cfg=dict(type='Conv3d'),
                 act_cfg=dict(type='ReLU'), with_cp=False, upsample_cfg=dict(mode='trilinear'), init_cfg=None,
                 compute_dtype=torch.float32):
        super().__init__()
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self.with_cp, self.upsample_cfg, self.compute_dtype = with_cp, dict(upsample_cfg), compute_dtype
        self.num_out = len(self.in_channels)
        mk = lambda cin, k, p: nn.Sequential(ConvModule(cin, out_channels, k, padding=p, conv_cfg=conv_cfg,  # noqa: E731
                                                        norm_cfg=norm_cfg, act_cfg=act_cfg, bias=False, inplace=True))
        self.lateral_convs = nn.ModuleList(mk(c, 1, 0) for c in self.in_channels)
        self.fpn_convs = nn.ModuleList(mk(out_channels, 3, 1) for _ in self.in_channels)

    def _run(self, mod, x):
        return checkpoint(mod, x, use_reentrant=False) if (self.with_cp and x.requires_grad) else mod(x)

    def _forward(self, inputs):
        laterals = [self._run(conv, x) for conv, x in zip(self.lateral_convs, inputs)]
        for i in range(self.num_out - 1, 0, -1):                                       # :92-96
            if self.upsample_cfg.get('mode') == 'trilinear' and len(self.upsample_cfg) == 1:
                up = upsample_trilinear(laterals[i], laterals[i - 1].shape[2:])       # separable form under autograd
            else:
                up = F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], align_corners=False, **self.upsample_cfg)
            laterals[i - 1] = laterals[i - 1] + up
        return [self._run(conv, x) for conv, x in zip(self.fpn_convs, laterals)]

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        if _low_precision_ok(self.compute_dtype, inputs[0]):
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._forward(inputs)
        return self._forward([x.float() for x in inputs])
