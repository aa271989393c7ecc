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


class FoldedConv3d:
    """One launch of fbbev_conv3d_ndhwc: conv (+ folded BN) (+ residual) (+ ReLU) on NDHWC activations."""

    def __init__(self, conv, bn=None, relu=False):
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
        self.wf = weight_fragments(w, self.transposed)
        self.bias = F.pad(b, (0, (self.cout + 15) // 16 * 16 - self.cout)).contiguous()

    def out_shape(self, x):
        B, D, H, W, _ = x.shape
        if self.transposed:
            return (B, 2 * D, 2 * H, 2 * W, self.cout)
        f = lambda n: (n + 2 * self.pad - self.ksize) // self.stride + 1  # noqa: E731
        return (B, f(D), f(H), f(W), self.cout)

    def __call__(self, x, residual=None, backend=None):
        out = torch.empty(self.out_shape(x), dtype=torch.float32, device=x.device)
        run = backend or _capi.conv3d_ndhwc
        return run(x, self.wf, self.bias, out, self.cout, ksize=self.ksize, stride=self.stride, pad=self.pad, relu=self.relu,
                   residual=residual, transposed=self.transposed)


def to_ndhwc(x):
    """(B,C,D,H,W) logical -> (B,D,H,W,C) contiguous (free when x is channels_last_3d)."""
    return x.permute(0, 2, 3, 4, 1).contiguous().float()


def to_ncdhw(x):
    """(B,D,H,W,C) contiguous -> (B,C,D,H,W) logical view (channels_last_3d memory format), no copy."""
    return x.permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------ the three stacks
class ResNet3DRunner:
    """CustomResNet3D.forward (resnet3d.py:248-274) with every conv+BN(+residual)+ReLU group as one launch."""

    def __init__(self, net):
        if net.plane2voxel is not None:
            raise NotImplementedError('plane2voxel')
        self.out_indices = net.out_indices
        self.input_proj = FoldedConv3d(net.input_proj[0], net.input_proj[1], relu=True)
        self.stages = []
        for layer in net.layers:
            blocks = []
            for blk in layer:
                down = None if blk.downsample is None else FoldedConv3d(blk.downsample[0], blk.downsample[1], relu=False)
                blocks.append((FoldedConv3d(blk.conv1, blk.bn1, relu=True), FoldedConv3d(blk.conv2, blk.bn2, relu=True), down))
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

    def __init__(self, neck):
        mk = lambda seq: FoldedConv3d(seq[0].conv, getattr(seq[0], seq[0].norm_name), relu=seq[0].activate is not None)  # noqa: E731
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

    def __init__(self, head):
        self.deblock = FoldedConv3d(head.deblock[0], head.deblock[1], relu=True) if head.use_deblock else None
        self.occ_convs = [FoldedConv3d(s[0], s[1], relu=True) for s in head.occ_convs]
        self.pred = (FoldedConv3d(head.occ_pred_conv[0], head.occ_pred_conv[1], relu=True),
                     FoldedConv3d(head.occ_pred_conv[3], None, relu=False))
        self.soft = None
        if head.soft_weights:
            self.soft = (FoldedConv3d(head.voxel_soft_weights[0], head.voxel_soft_weights[1], relu=True),
                         FoldedConv3d(head.voxel_soft_weights[3], None, relu=False))
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
