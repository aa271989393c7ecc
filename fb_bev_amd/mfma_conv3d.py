"""fp32-MFMA execution of FB-OCC's 3-D convolution stacks for inference: host side of
`fbbev_conv3d_ndhwc` (csrc/conv3d_kernels.h).

The reference runs CustomResNet3D (resnet3d.py:143-274), FPN3D (fpn3d.py:14-110) and OccHead
(occupancy_head.py:143-181) in fp32 through the vendor convolution library; at the shipped config those stacks are
670 GFLOP per frame and 33 of the 41.7 ms a frame takes on MI355X (profiles/r01_time_full.jsonl).  This module maps the
eval-mode modules onto the HIP kernel: every Conv3d + BatchNorm3d (+ residual) (+ ReLU) group becomes ONE launch with
the batch norm folded into weights and bias, activations stay NDHWC between launches, and the interpolation / softmax
blend steps in between remain torch ops on channels_last_3d views of the same buffers.

STATUS: the kernel and this mapping are validated on the CPU device emulator (tests/test_emu_conv3d.py) against
torch's fp32 convolutions; they have NOT run on the GPU yet (the round's GPU budget ended first), so the detector keeps
the vendor path unless `execution=dict(mfma_conv3d=True)` is passed.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi


# ------------------------------------------------------------------ weight layout
def weight_fragments(w, transposed=False):
    """Conv3d weight (Cout, Cin, k, k, k) [ConvTranspose3d k=2 s=2: (Cin, Cout, 2, 2, 2)] -> flat f32 tensor in the
    A-fragment order of fbbev_conv3d_ndhwc:  wf[parity][tap][j][mt][lane = 16 kk + i][e] =
    W[cout = 16 mt + i][cin = 16 j + 4 kk + e][tap], zero for cout >= Cout."""
    if transposed:
        assert tuple(w.shape[2:]) == (2, 2, 2)
        blocks = [weight_fragments(w[:, :, a, b, c].transpose(0, 1).reshape(w.shape[1], w.shape[0], 1, 1, 1))
                  for a in range(2) for b in range(2) for c in range(2)]
        return torch.cat(blocks)
    Cout, Cin = w.shape[:2]
    T = w[0, 0].numel()
    if Cin % 16:
        raise ValueError('fbbev_conv3d_ndhwc needs Cin % 16 == 0')
    MT = (Cout + 15) // 16
    wp = F.pad(w.reshape(Cout, Cin, T).float(), (0, 0, 0, 0, 0, 16 * MT - Cout))            # (16 MT, Cin, T)
    wp = wp.view(MT, 16, Cin // 16, 4, 4, T)                                                # mt, i, j, kk, e, tap
    return wp.permute(5, 2, 0, 3, 1, 4).contiguous().view(-1)                               # tap, j, mt, kk, i, e


def weight_fragments_bf16(w, transposed=False):
    """bf16 fragment layout of fbbev_conv3d_ndhwc_bf16: wfb[parity][tap][j][mt][lane = 16 g + i][e] =
    bf16(W[cout = 16 mt + i][cin = 32 j + 8 g + e][tap]) -> flat torch.bfloat16 tensor."""
    if transposed:
        blocks = [weight_fragments_bf16(w[:, :, a, b, c].transpose(0, 1).reshape(w.shape[1], w.shape[0], 1, 1, 1))
                  for a in range(2) for b in range(2) for c in range(2)]
        return torch.cat(blocks)
    Cout, Cin = w.shape[:2]
    T = w[0, 0].numel()
    if Cin % 32:
        raise ValueError('fbbev_conv3d_ndhwc_bf16 needs Cin % 32 == 0')
    MT = (Cout + 15) // 16
    wp = F.pad(w.reshape(Cout, Cin, T).float(), (0, 0, 0, 0, 0, 16 * MT - Cout))            # (16 MT, Cin, T)
    wp = wp.view(MT, 16, Cin // 32, 4, 8, T)                                                # mt, i, j, g, e, tap
    return wp.permute(5, 2, 0, 3, 1, 4).contiguous().view(-1).to(torch.bfloat16)            # tap, j, mt, g, i, e


def fold(conv, bn=None):
    """(weight, bias) of conv followed by eval-mode batch norm as one affine convolution."""
    w = conv.weight.detach().float()
    transposed = isinstance(conv, nn.ConvTranspose3d)
    cout = w.shape[1] if transposed else w.shape[0]
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout, device=w.device)
    if bn is not None:
        if not isinstance(bn, nn.modules.batchnorm._BatchNorm):
            raise NotImplementedError('only BatchNorm folds into a convolution (GroupNorm depends on the activations)')
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
        shape = (1, -1, 1, 1, 1) if transposed else (-1, 1, 1, 1, 1)
        w = w * scale.view(shape)
        b = (b - bn.running_mean.float()) * scale + bn.bias.detach().float()
    return w, b


def _launch(x, wf, bias, out, cout, backend=None, planar=False, tiled=False, **kw):
    """One convolution launch.  The kernel follows the weight layout: bf16 fragments -> fbbev_conv3d_ndhwc_bf16, fp32
    fragments -> fbbev_conv3d_ndhwc / fbbev_conv2d_nhwc.  `backend` (tests) stands in for the HIP entry points."""
    if backend is not None:
        return backend(x, wf, bias, out, cout, planar=planar, tiled=tiled, **kw)
    if wf.dtype == torch.bfloat16:
        if tiled:                                   # 3x3x3 / stride 1 / padding 1: LDS-staged halo tile
            return _capi.conv3d_k3s1_tiled_bf16(x, wf, bias, out, cout, relu=kw.get('relu', False), residual=kw.get('residual'))
        if planar:
            _capi.conv3d_ndhwc_bf16(x.unsqueeze(1), wf, bias, out.unsqueeze(1), cout, planar=True, **kw)
            return out
        return _capi.conv3d_ndhwc_bf16(x, wf, bias, out, cout, **kw)
    if planar:
        return _capi.conv2d_nhwc(x, wf, bias, out, cout, **kw)
    return _capi.conv3d_ndhwc(x, wf, bias, out, cout, **kw)


class FoldedConv3d:
    """One launch of fbbev_conv3d_ndhwc[_bf16]: conv (+ folded BN) (+ residual) (+ ReLU) on NDHWC activations.
    precision='bf16' takes the bf16-MFMA kernel where the input channels allow it (Cin % 32 == 0), fp32 otherwise;
    'bf16_tiled' additionally routes 3x3x3 / stride 1 / padding 1 layers to the LDS-halo kernel."""

    def __init__(self, conv, bn=None, relu=False, precision='f32'):
        self.transposed = isinstance(conv, nn.ConvTranspose3d)
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        if len(set(k)) != 1 or len(set(s)) != 1 or len(set(p)) != 1:
            raise NotImplementedError('anisotropic kernel / stride / padding')
        if self.transposed:
            if (k[0], s[0], p[0]) != (2, 2, 0):
                raise NotImplementedError('only ConvTranspose3d(kernel 2, stride 2, padding 0)')
        elif k[0] not in (1, 3) or s[0] not in (1, 2) or p[0] not in (0, 1) or conv.groups != 1 or set(conv.dilation) != {1}:
            raise NotImplementedError(f'conv {k} stride {s} pad {p}')
        self.ksize, self.stride, self.pad, self.relu = k[0], s[0], p[0], relu
        w, b = fold(conv, bn)
        self.cout = w.shape[1] if self.transposed else w.shape[0]
        cin = w.shape[0] if self.transposed else w.shape[1]
        bf16 = precision in ('bf16', 'bf16_tiled') and cin % 32 == 0
        self.wf = weight_fragments_bf16(w, self.transposed) if bf16 else weight_fragments(w, self.transposed)
        self.tiled = bf16 and precision == 'bf16_tiled' and not self.transposed and (self.ksize, self.stride, self.pad) == (3, 1, 1)
        self.bias = F.pad(b, (0, (self.cout + 15) // 16 * 16 - self.cout)).contiguous()

    def out_shape(self, x):
        B, D, H, W, _ = x.shape
        if self.transposed:
            return (B, 2 * D, 2 * H, 2 * W, self.cout)
        f = lambda n: (n + 2 * self.pad - self.ksize) // self.stride + 1  # noqa: E731
        return (B, f(D), f(H), f(W), self.cout)

    def __call__(self, x, residual=None, backend=None):
        out = torch.empty(self.out_shape(x), dtype=torch.float32, device=x.device)
        return _launch(x, self.wf, self.bias, out, self.cout, backend=backend, tiled=self.tiled, ksize=self.ksize,
                       stride=self.stride, pad=self.pad, relu=self.relu, residual=residual, transposed=self.transposed)


# ------------------------------------------------------------------ training route (autograd)
def _supported_train(conv, x):
    if isinstance(conv, nn.ConvTranspose3d):
        cin, cout = conv.in_channels, conv.out_channels
        ok = (conv.kernel_size, conv.stride, conv.padding, conv.output_padding) == ((2, 2, 2), (2, 2, 2), (0, 0, 0), (0, 0, 0))
    else:
        cin, cout = conv.in_channels, conv.out_channels
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        ok = len(set(k)) == len(set(s)) == len(set(p)) == 1 and k[0] in (1, 3) and s[0] in (1, 2) and p[0] in (0, 1) \
            and set(conv.dilation) == {1} and conv.padding_mode == 'zeros'
    if isinstance(conv, nn.ConvTranspose3d):
        return ok and conv.groups == 1 and cin % 16 == 0 and cout % 16 == 0 and x.dtype == torch.float32
    # plain convolutions take any channel count: MConv3d pads Cin / Cout to the kernels' multiple of 16 with zeros
    return ok and conv.groups == 1 and x.dtype == torch.float32


class _Conv3dFn(torch.autograd.Function):
    """Conv3d on NDHWC f32: forward fbbev_conv3d_ndhwc, backward fbbev_conv3d_dgrad_ndhwc + fbbev_conv3d_wgrad_ndhwc."""

    @staticmethod
    def forward(ctx, x, weight, bias, ksize, stride, pad, backends):
        fwd = backends[0] if backends else _capi.conv3d_ndhwc
        cout = weight.shape[0]
        b = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=x.device)
        if bias is not None:
            b[:cout] = bias.detach()
        B, D, H, W, _ = x.shape
        f = lambda n: (n + 2 * pad - ksize) // stride + 1  # noqa: E731
        out = torch.empty((B, f(D), f(H), f(W), cout), dtype=torch.float32, device=x.device)
        out = fwd(x, weight_fragments(weight.detach()), b, out, cout, ksize=ksize, stride=stride, pad=pad)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (ksize, stride, pad, bias is not None, backends)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ksize, stride, pad, has_bias, backends = ctx.cfg
        dgrad = backends[1] if backends else _capi.conv3d_dgrad_ndhwc
        wgrad = backends[2] if backends else _capi.conv3d_wgrad_ndhwc
        dy = dy.contiguous().float()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(dy, weight_fragments(weight.detach().transpose(0, 1)), torch.empty_like(x), ksize=ksize, stride=stride, pad=pad)
        if ctx.needs_input_grad[1]:
            cout, cin = weight.shape[:2]
            dwt = wgrad(x, dy, torch.zeros((ksize ** 3, cout, cin), dtype=torch.float32, device=x.device), ksize=ksize,
                        stride=stride, pad=pad)
            dw = dwt.view(ksize, ksize, ksize, cout, cin).permute(3, 4, 0, 1, 2).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 1, 2, 3))
        return dx, dw, db, None, None, None, None


class _ConvTranspose3dFn(torch.autograd.Function):
    """ConvTranspose3d(k=2, s=2, p=0) on NDHWC f32.  Its data gradient is a kernel-2 stride-2 convolution of dy with the
    same weight read as (out=Cin, in=Cout, 2,2,2); its weight gradient the wgrad of that convolution (x := dy fine,
    dy := x coarse)."""

    @staticmethod
    def forward(ctx, x, weight, bias, backends):
        fwd = backends[0] if backends else _capi.conv3d_ndhwc
        cout = weight.shape[1]
        b = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=x.device)
        if bias is not None:
            b[:cout] = bias.detach()
        B, D, H, W, _ = x.shape
        out = torch.empty((B, 2 * D, 2 * H, 2 * W, cout), dtype=torch.float32, device=x.device)
        out = fwd(x, weight_fragments(weight.detach(), transposed=True), b, out, cout, transposed=True)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (bias is not None, backends)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        has_bias, backends = ctx.cfg
        fwd = backends[0] if backends else _capi.conv3d_ndhwc
        wgrad = backends[2] if backends else _capi.conv3d_wgrad_ndhwc
        dy = dy.contiguous().float()
        cin, cout = weight.shape[:2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            zero = torch.zeros((cin + 15) // 16 * 16, dtype=torch.float32, device=x.device)
            dx = fwd(dy, weight_fragments(weight.detach()), zero, torch.empty_like(x), cin, ksize=2, stride=2, pad=0)
        if ctx.needs_input_grad[1]:
            dwt = wgrad(dy, x, torch.zeros((8, cin, cout), dtype=torch.float32, device=x.device), ksize=2, stride=2, pad=0)
            dw = dwt.view(2, 2, 2, cin, cout).permute(3, 4, 0, 1, 2).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 1, 2, 3))
        return dx, dw, db, None


class MConv3d(nn.Conv3d):
    """nn.Conv3d whose forward can take the fp32-MFMA route under autograd (`mfma = True`, set by
    enable_training_route); parameters, state-dict names and the default behaviour are nn.Conv3d's."""
    mfma = False
    backends = None            # tests: (forward, dgrad, wgrad) callables standing in for the HIP entry points

    def forward(self, x):
        if self.mfma and (x.is_cuda or self.backends) and not torch.is_autocast_enabled() and _supported_train(self, x):
            # Channel counts off the kernels' multiple of 16 (the head's 19-class / 4-weight outputs, the history fusion's
            # 81 inputs) are zero-padded here: on this ROCm image the vendor library runs exactly those layers with its
            # naive fallback kernels (0.3-0.5 s EACH per training step at 200x200x16, profiles/r02_rocprofv3_train_step.csv).
            cin, cout = self.in_channels, self.out_channels
            pi, po = (-cin) % 16, (-cout) % 16
            xn, w, b = to_ndhwc(x), self.weight, self.bias
            if pi:
                xn, w = F.pad(xn, (0, pi)), F.pad(w, (0, 0, 0, 0, 0, 0, 0, pi))
            if po:
                w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, 0, 0, po))
                b = F.pad(b, (0, po)) if b is not None else None
            y = _Conv3dFn.apply(xn, w, b, self.kernel_size[0], self.stride[0], self.padding[0], self.backends)
            return to_ncdhw(y[..., :cout] if po else y)
        return super().forward(x)


class MConvTranspose3d(nn.ConvTranspose3d):
    mfma = False
    backends = None

    def forward(self, x, output_size=None):
        if self.mfma and (x.is_cuda or self.backends) and not torch.is_autocast_enabled() and _supported_train(self, x):
            return to_ncdhw(_ConvTranspose3dFn.apply(to_ndhwc(x), self.weight, self.bias, self.backends))
        return super().forward(x, output_size)


def enable_training_route(module, on=True, backends=None):
    """Switch every supported 3-D convolution below `module` to the MFMA autograd route (or back)."""
    n = 0
    for m in module.modules():
        if isinstance(m, (MConv3d, MConvTranspose3d)):
            m.mfma, m.backends = bool(on), backends
            n += 1
    return n


def to_ndhwc(x):
    """(B,C,D,H,W) logical -> (B,D,H,W,C) contiguous (free when x is channels_last_3d)."""
    return x.permute(0, 2, 3, 4, 1).contiguous().float()


def to_ncdhw(x):
    """(B,D,H,W,C) contiguous -> (B,C,D,H,W) logical view (channels_last_3d memory format), no copy."""
    return x.permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------ the three stacks
class ResNet3DRunner:
    """CustomResNet3D.forward (resnet3d.py:248-274) with every conv+BN(+residual)+ReLU group as one launch."""

    def __init__(self, net, precision='f32'):
        if net.plane2voxel is not None:
            raise NotImplementedError('plane2voxel')
        self.out_indices = net.out_indices
        FC = lambda *a, **k: FoldedConv3d(*a, precision=precision, **k)  # noqa: E731
        self.input_proj = FC(net.input_proj[0], net.input_proj[1], relu=True)
        self.stages = []
        for layer in net.layers:
            blocks = []
            for blk in layer:
                down = None if blk.downsample is None else FC(blk.downsample[0], blk.downsample[1], relu=False)
                blocks.append((FC(blk.conv1, blk.bn1, relu=True), FC(blk.conv2, blk.bn2, relu=True), down))
            self.stages.append(blocks)

    def __call__(self, x, backend=None):
        """x (B,D,H,W,C) NDHWC -> list of NDHWC feature maps."""
        x = self.input_proj(x, backend=backend)
        res = []
        for i, blocks in enumerate(self.stages):
            for c1, c2, down in blocks:
                identity = x if down is None else down(x, backend=backend)
                x = c2(c1(x, backend=backend), residual=identity, backend=backend)        # relu(bn2(conv2(.)) + identity)
            if i in self.out_indices:
                res.append(x)
        return res


class FPN3DRunner:
    """FPN3D.forward (fpn3d.py:72-110)."""

    def __init__(self, neck, precision='f32'):
        mk = lambda seq: FoldedConv3d(seq[0].conv, getattr(seq[0], seq[0].norm_name), relu=seq[0].activate is not None,  # noqa: E731
                                      precision=precision)
        self.laterals = [mk(s) for s in neck.lateral_convs]
        self.outs = [mk(s) for s in neck.fpn_convs]
        self.upsample_cfg = dict(neck.upsample_cfg)

    def __call__(self, feats, backend=None):
        lat = [conv(x, backend=backend) for conv, x in zip(self.laterals, feats)]
        for i in range(len(lat) - 1, 0, -1):
            up = F.interpolate(to_ncdhw(lat[i]), size=lat[i - 1].shape[1:4], align_corners=False, **self.upsample_cfg)
            lat[i - 1] = lat[i - 1] + up.permute(0, 2, 3, 4, 1)
        return [conv(x.contiguous(), backend=backend) for conv, x in zip(self.outs, lat)]


class OccHeadRunner:
    """OccHead.forward_coarse_voxel (occupancy_head.py:143-181) -> class logits (B, classes, H, W, D) like the module."""

    def __init__(self, head, precision='f32'):
        FC = lambda *a, **k: FoldedConv3d(*a, precision=precision, **k)  # noqa: E731
        self.deblock = FC(head.deblock[0], head.deblock[1], relu=True) if head.use_deblock else None
        self.occ_convs = [FC(s[0], s[1], relu=True) for s in head.occ_convs]
        self.pred = (FC(head.occ_pred_conv[0], head.occ_pred_conv[1], relu=True), FC(head.occ_pred_conv[3], None, relu=False))
        self.soft = None
        if head.soft_weights:
            self.soft = (FC(head.voxel_soft_weights[0], head.voxel_soft_weights[1], relu=True),
                         FC(head.voxel_soft_weights[3], None, relu=False))
        self.n_feat = head.num_point_sampling_feat

    def __call__(self, feats, backend=None, blend_backend=None):
        occs = []
        if self.deblock is not None:
            occs.append(self.deblock(feats[0], backend=backend))
        occs += [conv(x, backend=backend) for conv, x in zip(self.occ_convs, feats)]
        size = occs[0].shape[1:4]
        if self.soft is not None:
            w = torch.softmax(self.soft[1](self.soft[0](occs[0], backend=backend), backend=backend), dim=-1)   # (B,D,H,W,n)
        else:
            w = occs[0].new_full((*occs[0].shape[:4], self.n_feat), 1.0 / self.n_feat)
        if len(occs) <= 4 and occs[0].shape[-1] % 4 == 0:
            # one pass: level 0 read once, the coarse levels sampled in the kernel, the blended map written once
            blend = blend_backend or _capi.blend_levels_ndhwc
            out = blend(occs[0], [f.contiguous() for f in occs[1:]], w.contiguous(), torch.empty_like(occs[0]))
        else:
            out = 0
            for k, f in enumerate(occs):
                if tuple(f.shape[1:4]) != tuple(size):
                    f = F.interpolate(to_ncdhw(f), size=list(size), mode='trilinear', align_corners=False).permute(0, 2, 3, 4, 1)
                out = out + f * w[..., k:k + 1]
        logits = self.pred[1](self.pred[0](out.contiguous(), backend=backend), backend=backend)
        return to_ncdhw(logits)


# ------------------------------------------------------------------ 2-D stacks (image encoder), inference
class FoldedConv2d:
    """One launch of fbbev_conv2d_nhwc: Conv2d (+ folded BN) (+ residual) (+ ReLU) on NHWC activations."""

    def __init__(self, conv, bn=None, relu=False, precision='f32'):
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        if len(set(k)) != 1 or len(set(s)) != 1 or len(set(p)) != 1 or k[0] not in (1, 3) or s[0] not in (1, 2) \
                or p[0] not in (0, 1) or conv.groups != 1 or set(conv.dilation) != {1}:
            raise NotImplementedError(f'conv2d {k} stride {s} pad {p}')
        self.ksize, self.stride, self.pad, self.relu = k[0], s[0], p[0], relu
        w = conv.weight.detach().float()
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if bn is not None:
            scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w = w * scale.view(-1, 1, 1, 1)
            b = (b - bn.running_mean.float()) * scale + bn.bias.detach().float()
        self.cout = w.shape[0]
        self.wf = weight_fragments_bf16(w[:, :, None]) if (precision in ('bf16', 'bf16_tiled') and w.shape[1] % 32 == 0) \
            else weight_fragments(w[:, :, None])
        self.bias = F.pad(b, (0, (self.cout + 15) // 16 * 16 - self.cout)).contiguous()

    def __call__(self, x, residual=None, backend=None):
        B, H, W, _ = x.shape
        f = lambda n: (n + 2 * self.pad - self.ksize) // self.stride + 1  # noqa: E731
        out = torch.empty((B, f(H), f(W), self.cout), dtype=torch.float32, device=x.device)
        return _launch(x, self.wf, self.bias, out, self.cout, backend=backend, planar=True, ksize=self.ksize, stride=self.stride,
                       pad=self.pad, relu=self.relu, residual=residual)


class ResNetRunner:
    """img_encoder.ResNet.forward in eval mode: the 7x7 stem (3 input channels) and the max-pool stay torch; every
    residual block is 2-3 launches (conv+BN+ReLU ..., the last one with the identity added before the ReLU)."""

    def __init__(self, net, precision='f32'):
        self.net = net
        FC2 = lambda *a, **k: FoldedConv2d(*a, precision=precision, **k)  # noqa: E731
        self.out_indices = net.out_indices
        self.stages = []
        for name in net.res_layers:
            blocks = []
            for blk in getattr(net, name):
                down = None if blk.downsample is None else FC2(blk.downsample[0], blk.downsample[1], relu=False)
                if hasattr(blk, 'conv3'):
                    convs = [FC2(blk.conv1, blk.bn1, relu=True), FC2(blk.conv2, blk.bn2, relu=True), FC2(blk.conv3, blk.bn3, relu=True)]
                else:
                    convs = [FC2(blk.conv1, blk.bn1, relu=True), FC2(blk.conv2, blk.bn2, relu=True)]
                blocks.append((convs, down))
            self.stages.append(blocks)

    def __call__(self, img, backend=None):
        """img (B,3,H,W) -> tuple of NHWC feature maps of the out_indices stages."""
        net = self.net
        x = net.maxpool(F.relu(net.bn1(net.conv1(img.float()))))
        x = x.permute(0, 2, 3, 1).contiguous()
        outs = []
        for i, blocks in enumerate(self.stages):
            for convs, down in blocks:
                identity = x if down is None else down(x, backend=backend)
                y = x
                for c in convs[:-1]:
                    y = c(y, backend=backend)
                x = convs[-1](y, residual=identity, backend=backend)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class CustomFPNRunner:
    """img_encoder.CustomFPN.forward (necks/fpn.py:160-206) on NHWC maps -> (B,C,H,W) logical view of outs[0]."""

    def __init__(self, neck, precision='f32'):
        if len(neck.fpn_convs) != len(neck.out_ids) or neck.num_outs != len(neck.out_ids):
            raise NotImplementedError('extra output levels')
        mk = lambda m: FoldedConv2d(m.conv, getattr(m, m.norm_name) if m.norm_name else None, relu=m.activate is not None,  # noqa: E731
                                    precision=precision)
        self.laterals = [mk(m) for m in neck.lateral_convs]
        self.outs = [mk(m) for m in neck.fpn_convs]
        self.out_ids, self.start_level, self.upsample_cfg = list(neck.out_ids), neck.start_level, dict(neck.upsample_cfg)

    def __call__(self, feats, backend=None):
        lat = [conv(feats[i + self.start_level], backend=backend) for i, conv in enumerate(self.laterals)]
        for i in range(len(lat) - 1, 0, -1):
            up = F.interpolate(lat[i].permute(0, 3, 1, 2), size=lat[i - 1].shape[1:3], **self.upsample_cfg)
            lat[i - 1] = lat[i - 1] + up.permute(0, 2, 3, 1)
        return self.outs[0](lat[self.out_ids[0]].contiguous(), backend=backend).permute(0, 3, 1, 2)
