"""Image encoder in front of the depth net: `ResNet` backbone + `CustomFPN` neck.

Reference: FBOCC.image_encoder (mmdet3d/models/fbbev/detectors/fbocc.py:135-149) flattens the (B, N) camera axes, runs
`img_backbone` (mmdet `ResNet`, external to the tree: depth 50, out_indices (2, 3), style 'pytorch', cfg
occupancy_configs/fb_occ/fbocc-r50-cbgs_depth_16f_16x4_20e.py:119-129) and `img_neck` (`CustomFPN`,
mmdet3d/models/necks/fpn.py:12-206, cfg :130-137) and reshapes back to (B, N, C, H/16, W/16).

The ResNet is restated from its published architecture (He et al. 2016; torchvision / mmdet 'pytorch' style: stride on
the 3x3 conv of the bottleneck): state-dict names conv1 / bn1 / layer<k>.<i>.{conv1..3, bn1..3, downsample.{0,1}} equal
those of the `resnet50-0676ba61.pth` checkpoint the config names.  mmdet's ResNet is not in the reference tree, so
this block is "parity unpinned" against it; CustomFPN is pinned on a fixture from the real class.

Dense 2-D convolutions: MFMA-bound, on MIOpen.  Execution setup for MI355X: channels_last activations, optional bf16
compute dtype, the (B*N) camera batch in one call.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .bev_encoder import ConvModule, build_norm


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_cfg=None, style='pytorch'):
        super().__init__()
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = build_norm(norm_cfg, planes, 2)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=1, bias=False)
        self.bn2 = build_norm(norm_cfg, planes, 2)[1]
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = build_norm(norm_cfg, planes * 4, 2)[1]
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)


class BasicBlock2D(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_cfg=None, style='pytorch'):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = build_norm(norm_cfg, planes, 2)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = build_norm(norm_cfg, planes, 2)[1]
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)


class ResNet(nn.Module):
    """mmdet.models.backbones.ResNet (external) for the arguments the fb_occ configs pass."""
    arch = {18: (BasicBlock2D, (2, 2, 2, 2)), 34: (BasicBlock2D, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
            101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth=50, in_channels=3, base_channels=64, num_stages=4, strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 style='pytorch', frozen_stages=-1, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                 with_cp=False, pretrained=None, init_cfg=None, channels_last=True, compute_dtype=torch.float32):
        super().__init__()
        block, counts = self.arch[depth]
        self.out_indices, self.with_cp, self.norm_eval, self.frozen_stages = tuple(out_indices), with_cp, norm_eval, frozen_stages
        self.channels_last, self.compute_dtype = channels_last, compute_dtype
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = build_norm(norm_cfg, base_channels, 2)[1]
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = base_channels
        self.res_layers = []
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            downsample = None
            if strides[i] != 1 or inplanes != planes * block.expansion:
                downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=strides[i], bias=False),
                                           build_norm(norm_cfg, planes * block.expansion, 2)[1])
            blocks = [block(inplanes, planes, strides[i], downsample, norm_cfg, style)]
            inplanes = planes * block.expansion
            blocks += [block(inplanes, planes, norm_cfg=norm_cfg, style=style) for _ in range(1, counts[i])]
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def _forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            layer = getattr(self, name)
            if self.with_cp and x.requires_grad:
                for blk in layer:
                    x = checkpoint(blk, x, use_reentrant=False)
            else:
                x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, x):
        if self.channels_last and x.is_cuda:        # MIOpen layout; ATen's CPU NHWC backward is not relied on
            x = x.contiguous(memory_format=torch.channels_last)
        if self.compute_dtype != torch.float32 and x.is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._forward(x)
        return self._forward(x)


class CustomFPN(nn.Module):
    """mmdet3d/models/necks/fpn.py:12-206: lateral 1x1 convs on inputs[start_level:end], top-down nearest-neighbour
    merge (:171-179), 3x3 output convs only for the levels in `out_ids` (:123-134), returns outs[0] (:206)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, out_ids=(), add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, with_cp=False,
                 act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=None, compute_dtype=torch.float32):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        if add_extra_convs:
            raise NotImplementedError('add_extra_convs (:136-154) is not used by any fb_occ config')
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.upsample_cfg, self.out_ids, self.start_level = dict(upsample_cfg), list(out_ids), start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level
        self.compute_dtype = compute_dtype
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg, act_cfg=act_cfg,
                                                 inplace=False))
            if i in self.out_ids:
                self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))
        for m in self.modules():                        # init_cfg Xavier uniform on Conv2d (:82-83)
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _forward(self, inputs):
        laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            if 'scale_factor' in self.upsample_cfg:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], **self.upsample_cfg)
            else:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                                  **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in self.out_ids]
        if self.num_outs > len(outs):
            for _ in range(self.num_outs - len(laterals)):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return outs[0]

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        if self.compute_dtype != torch.float32 and inputs[0].is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._forward(inputs)
        return self._forward(inputs)
