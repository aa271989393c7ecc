"""Camera-aware depth net of FB-OCC: image features + camera parameters -> the two tensors the
lift-splat consumes, `context` (B,N,C,H,W) and the depth distribution `depth` (B,N,D,H,W).

Mirror of `CM_DepthNet` and its building blocks -- mmdet3d/models/fbbev/modules/depth_net.py:35-92 (`_ASPPModule`),
:95-174 (`ASPP`), :177-203 (`Mlp`), :206-220 (`SELayer`), :258-366 (`CM_DepthNet.__init__/forward`), :369-393
(`get_mlp_input`), :396-446 (depth supervision) -- and of mmdet's ResNet `BasicBlock` (external,
mmdet/models/backbones/resnet.py: conv3x3-BN-ReLU-conv3x3-BN + identity, ReLU) with the same sub-module names, so a
detector checkpoint's `depth_net.*` entries load unchanged.

This row is dense implicit GEMM (SURVEY: ~200 GFLOP per 6-camera sample) and therefore stays on the vendor libraries
(MIOpen convolutions, hipBLASLt linears) as BASELINE's north star prescribes ("MFMA used only for the dense
depth-net / value-projection GEMMs"); what is MI355X-specific here is the execution setup: channels-last activations
(MIOpen's NHWC kernels) and an optional bf16 autocast region around the convolution stack (`compute_dtype`), with the
softmax over depth bins and both outputs kept in fp32 for the fp32 lift-splat.  `use_dcn=True` (deformable conv,
mmcv CUDA op) is not built: the FB-OCC configs set `use_dcn=False`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_bn_relu(cin, cout, k, padding=0, dilation=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=padding, dilation=dilation, bias=False), \
        nn.BatchNorm2d(cout), nn.ReLU()


def _kaiming_unit_bn(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        self.atrous_conv, self.bn, self.relu = _conv_bn_relu(inplanes, planes, kernel_size, padding, dilation)
        _kaiming_unit_bn(self)

    def forward(self, x):
        return self.relu(self.bn(self.atrous_conv(x)))


class ASPP(nn.Module):
    """Four atrous branches (rates 1, 6, 12, 18) + image-level pooling, fused by a 1x1 conv (depth_net.py:95-174)."""

    def __init__(self, inplanes, mid_channels=256, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        for i, rate in enumerate((1, 6, 12, 18), start=1):
            k, pad = (1, 0) if rate == 1 else (3, rate)
            setattr(self, f'aspp{i}', _ASPPModule(inplanes, mid_channels, k, padding=pad, dilation=rate))
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                             nn.Conv2d(inplanes, mid_channels, 1, stride=1, bias=False),
                                             nn.BatchNorm2d(mid_channels), nn.ReLU())
        self.conv1 = nn.Conv2d(mid_channels * 5, mid_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid_channels)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        _kaiming_unit_bn(self)

    def forward(self, x):
        branches = [self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)]
        # depth_net.py:167-168: F.interpolate(1x1 -> HxW, bilinear, align_corners=True).  With a 1x1 source every output pixel has
        # source index 0 and interpolation weight (1, 0): the value itself, bit for bit -- a broadcast.  ATen's generic
        # upsample kernel took 2.0 ms of the 5.2 ms bf16 depth net at B = 4 x 6 images (38 % of its GPU time,
        # profiles/r04_pmc_mfma.json); its backward is the same sum over pixels that expand's backward is.
        pooled = self.global_avg_pool(x).expand(-1, -1, x.shape[2], x.shape[3])
        return self.dropout(self.relu(self.bn1(self.conv1(torch.cat(branches + [pooled], dim=1)))))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class SELayer(nn.Module):
    """Camera-parameter gate: x * sigmoid(expand(relu(reduce(x_se)))) (depth_net.py:206-220)."""

    def __init__(self, channels, act_layer=nn.ReLU, gate_layer=nn.Sigmoid):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = act_layer()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = gate_layer()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class BasicBlock(nn.Module):
    """mmdet ResNet BasicBlock (stride 1, no downsample branch -- the way depth_net.py:300-305 builds it)."""

    def __init__(self, inplanes, planes, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + identity)


class CM_DepthNet(nn.Module):
    """Constructor arguments of the reference (depth_net.py:262-276) + two execution knobs:
    channels_last (default True): run the convolution stack on NHWC activations;
    compute_dtype (default torch.float32, as the reference's @force_fp32): torch.bfloat16 wraps the conv stack in
    autocast; softmax and outputs stay fp32."""

    def __init__(self, in_channels=512, context_channels=64, depth_channels=118, mid_channels=512, use_dcn=True,
                 downsample=16, grid_config=None, loss_depth_weight=3.0, with_cp=False, se_depth_map=False, sid=False,
                 bias=0.0, input_size=None, use_aspp=True, channels_last=True, compute_dtype=torch.float32):
        super().__init__()
        if use_dcn:
            raise NotImplementedError('use_dcn=True needs mmcv\'s deformable convolution; the FB-OCC configs use use_dcn=False')
        self.sid, self.with_cp, self.downsample, self.grid_config = sid, with_cp, downsample, grid_config
        self.loss_depth_weight, self.se_depth_map = loss_depth_weight, se_depth_map
        self.context_channels, self.depth_channels = context_channels, depth_channels
        self.channels_last, self.compute_dtype = channels_last, compute_dtype
        self.reduce_conv = nn.Sequential(nn.Conv2d(in_channels, mid_channels, kernel_size=3, stride=1, padding=1),
                                         nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.context_conv = nn.Conv2d(mid_channels, context_channels, kernel_size=1, stride=1, padding=0)
        self.bn = nn.BatchNorm1d(27)
        self.depth_mlp = Mlp(27, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        self.context_mlp = Mlp(27, mid_channels, mid_channels)
        self.context_se = SELayer(mid_channels)
        stack = [BasicBlock(mid_channels, mid_channels) for _ in range(3)]
        if use_aspp:
            stack.append(ASPP(mid_channels, mid_channels))
        stack.append(nn.Conv2d(mid_channels, depth_channels, kernel_size=1, stride=1, padding=0))
        self.depth_conv = nn.Sequential(*stack)

    # ------------------------------------------------------------------ forward (depth_net.py:335-366)
    def _trunk(self, x, cam):
        x = self.reduce_conv(x)
        context = self.context_conv(self.context_se(x, self.context_mlp(cam)[..., None, None]))
        depth = self.depth_conv(self.depth_se(x, self.depth_mlp(cam)[..., None, None]))
        return context, depth

    def forward(self, x, mlp_input):
        B, N, C, H, W = x.shape
        x = x.to(torch.float32).view(B * N, C, H, W)
        cam = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]).float())
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if self.compute_dtype != torch.float32 and x.is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                context, depth = self._trunk(x, cam)
        elif self.with_cp and x.requires_grad:
            context, depth = torch.utils.checkpoint.checkpoint(self._trunk, x, cam, use_reentrant=False)
        else:
            context, depth = self._trunk(x, cam)
        depth = depth.float().softmax(dim=1)
        context = context.float().contiguous().view(B, N, self.context_channels, H, W)
        return context, depth.contiguous().view(B, N, self.depth_channels, H, W)

    # ------------------------------------------------------------------ camera descriptor (depth_net.py:369-393)
    @staticmethod
    def get_mlp_input(rot, tran, intrin, post_rot, post_tran, bda):
        """27 numbers per camera: fx fy cx cy | the 2x3 image augmentation | 5 entries of the BEV augmentation |
        the 3x4 sensor-to-ego transform."""
        B, N = rot.shape[:2]
        bda = bda.view(B, 1, 3, 3).expand(B, N, 3, 3)
        head = torch.stack([intrin[..., 0, 0], intrin[..., 1, 1], intrin[..., 0, 2], intrin[..., 1, 2],
                            post_rot[..., 0, 0], post_rot[..., 0, 1], post_tran[..., 0],
                            post_rot[..., 1, 0], post_rot[..., 1, 1], post_tran[..., 1],
                            bda[..., 0, 0], bda[..., 0, 1], bda[..., 1, 0], bda[..., 1, 1], bda[..., 2, 2]], dim=-1)
        sensor2ego = torch.cat([rot, tran.reshape(B, N, 3, 1)], dim=-1).reshape(B, N, 12)
        return torch.cat([head, sensor2ego], dim=-1)

    # ------------------------------------------------------------------ depth supervision (depth_net.py:396-446)
    def get_downsampled_gt_depth(self, gt_depths):
        """(B,N,H,W) metric depth maps (0 = no return) -> one-hot bins (B*N*h*w, D): nearest non-zero depth of every
        downsample x downsample cell."""
        ds = self.downsample
        B, N, H, W = gt_depths.shape
        cells = gt_depths.view(B * N, H // ds, ds, W // ds, ds).permute(0, 1, 3, 2, 4).reshape(-1, ds * ds)
        nearest = torch.where(cells == 0.0, torch.full_like(cells, 1e5), cells).min(dim=-1).values
        nearest = nearest.view(B * N, H // ds, W // ds)
        lo, hi, step = self.grid_config['depth']
        if not self.sid:
            bins = (nearest - (lo - step)) / step
        else:
            bins = torch.log(nearest) - torch.log(torch.tensor(lo).float())
            bins = bins * (self.depth_channels - 1) / torch.log(torch.tensor(hi - 1.).float() / lo) + 1.
        bins = torch.where((bins < self.depth_channels + 1) & (bins >= 0.0), bins, torch.zeros_like(bins))
        onehot = F.one_hot(bins.long(), num_classes=self.depth_channels + 1).view(-1, self.depth_channels + 1)
        return onehot[:, 1:].float()

    def get_depth_loss(self, depth_labels, depth_preds):
        labels = self.get_downsampled_gt_depth(depth_labels)
        preds = depth_preds.permute(0, 1, 3, 4, 2).contiguous().view(-1, self.depth_channels)
        fg = labels.max(dim=1).values > 0.0
        with torch.autocast('cuda', enabled=False):
            # the reference compacts the foreground rows first (`preds[fg_mask]`, depth_net.py:441-448: a nonzero() + host
            # sync per step); multiplying by the mask sums the same terms without leaving the stream
            bce = F.binary_cross_entropy(preds.float(), labels, reduction='none')
            loss = (bce * fg[:, None].float()).sum() / fg.sum().clamp(min=1.0)
        return dict(loss_depth=self.loss_depth_weight * loss)
