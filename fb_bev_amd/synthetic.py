"""Seeded synthetic inputs for the view-transformation hot path.

The shapes and the camera rig follow SURVEY.md section 8(d): a deterministic nuScenes-like
6-camera rig (the reference builds the real thing in
mmdet3d/datasets/pipelines/loading.py:1078-1088,1186-1308), a peaky depth distribution
(softmax of 3*N(0,1) over D, like a trained CM_DepthNet output, depth_net.py:335-366) and
N(0,1) context features.  Everything is generated on CPU from a fixed seed so that the CPU
oracle and the GPU path see identical bits.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch


@dataclass(frozen=True)
class PathConfig:
    """One named workload of the path (names follow SURVEY.md section 8d / BASELINE.json)."""
    name: str
    input_size: Tuple[int, int]          # (H_in, W_in)
    downsample: int
    grid_config: Dict[str, List[float]]  # x, y, z, depth : [lower, upper, interval]
    channels: int                        # C = numC_Trans
    n_cams: int = 6

    @property
    def feat_hw(self):
        return self.input_size[0] // self.downsample, self.input_size[1] // self.downsample

    @property
    def D(self):
        lo, hi, st = self.grid_config['depth']
        return int(torch.arange(lo, hi, st, dtype=torch.float).shape[0])

    @property
    def grid_xyz(self):
        # same float arithmetic as view_transformer.py:386-387 (python float, then fp32 tensor)
        gs = torch.Tensor([(c[1] - c[0]) / c[2] for c in
                           (self.grid_config['x'], self.grid_config['y'], self.grid_config['z'])])
        return int(gs[0]), int(gs[1]), int(gs[2])


def _grid(x, y, z, depth):
    return {'x': list(x), 'y': list(y), 'z': list(z), 'depth': list(depth)}


CONFIGS = {
    # shipped config: occupancy_configs/fb_occ/fbocc-r50-cbgs_depth_16f_16x4_20e.py:56,78-94
    'REF': PathConfig('REF', (256, 704), 16,
                      _grid([-40, 40, 0.8], [-40, 40, 0.8], [-1, 5.4, 0.8], [2.0, 42.0, 0.5]), 80),
    # BASELINE.json configs[0]: 64x176 feat, D=59, 100x100x8 grid, C=64
    'BL1': PathConfig('BL1', (256, 704), 4,
                      _grid([-40, 40, 0.8], [-40, 40, 0.8], [-1, 5.4, 0.8], [1.0, 60.0, 1.0]), 64),
    # BASELINE.json configs[1]: 6x256x704, D=59, 200x200x16 BEV, C=80 (bench workload)
    'BL2': PathConfig('BL2', (256, 704), 16,
                      _grid([-40, 40, 0.4], [-40, 40, 0.4], [-1, 5.4, 0.4], [1.0, 60.0, 1.0]), 80),
    # BASELINE.json configs[4]: 6x512x1408, D=118, 400x400x16 grid (stress)
    'BL5': PathConfig('BL5', (512, 1408), 16,
                      _grid([-40, 40, 0.2], [-40, 40, 0.2], [-1, 5.4, 0.4], [1.0, 60.0, 0.5]), 80),
    # small cases for loop-exact oracles / golden fixtures
    'TINY': PathConfig('TINY', (32, 48), 8,
                       _grid([-8, 8, 1.0], [-8, 8, 1.0], [-1, 3, 1.0], [1.0, 9.0, 1.0]), 8, n_cams=6),
    'SMALL': PathConfig('SMALL', (64, 176), 8,
                        _grid([-20, 20, 0.8], [-20, 20, 0.8], [-1, 5.4, 0.8], [1.0, 30.0, 1.0]), 20),
}


def _rz(deg):
    a = math.radians(deg)
    return torch.tensor([[math.cos(a), -math.sin(a), 0.], [math.sin(a), math.cos(a), 0.],
                         [0., 0., 1.]], dtype=torch.float64)


def camera_rig(cfg: PathConfig, batch: int, seed: int = 0, bda_aug: bool = False):
    """cam_params = (rots, trans, intrins, post_rots, post_trans, bda), fp32, CPU.

    Mirrors the tuple FBOCC hands to the view transformers (fbocc.py:328 `cam_params = img[1:7]`).
    """
    H_in, W_in = cfg.input_size
    yaws = [55., 0., -55., 110., 180., -110.][:cfg.n_cams]
    cam2ego_axes = torch.tensor([[0., 0., 1.], [-1., 0., 0.], [0., -1., 0.]], dtype=torch.float64)
    rots, trans = [], []
    for yaw in yaws:
        rots.append(_rz(yaw) @ cam2ego_axes)
        a = math.radians(yaw)
        trans.append(torch.tensor([1.5 * math.cos(a), 0.5 * math.sin(a), 1.5], dtype=torch.float64))
    rots = torch.stack(rots).float()[None].repeat(batch, 1, 1, 1)
    trans = torch.stack(trans).float()[None].repeat(batch, 1, 1)
    K = torch.tensor([[1266., 0., 816.], [0., 1266., 491.], [0., 0., 1.]])
    intrins = K[None, None].repeat(batch, cfg.n_cams, 1, 1)
    s = W_in / 1600.0
    post_rots = torch.diag(torch.tensor([s, s, 1.0]))[None, None].repeat(batch, cfg.n_cams, 1, 1)
    post_trans = torch.tensor([0., -(int(900 * s) - H_in), 0.])[None, None].repeat(batch, cfg.n_cams, 1)
    bda = torch.eye(3)[None].repeat(batch, 1, 1)
    if bda_aug:  # training-time BEV augmentation (cfg :67-71): rotation +-22.5 deg, random flips
        g = torch.Generator().manual_seed(seed + 17)
        for b in range(batch):
            ang = (torch.rand(1, generator=g).item() - 0.5) * 45.0
            fx = -1.0 if torch.rand(1, generator=g).item() < 0.5 else 1.0
            fy = -1.0 if torch.rand(1, generator=g).item() < 0.5 else 1.0
            bda[b] = (_rz(ang) @ torch.diag(torch.tensor([fx, fy, 1.0], dtype=torch.float64))).float()
        # per-sample jitter of the rig so different samples rank differently
        jit = (torch.rand(batch, cfg.n_cams, 3, generator=g) - 0.5) * 0.2
        trans = trans + jit
    return tuple(t.contiguous() for t in (rots, trans, intrins, post_rots, post_trans, bda))


def depth_and_context(cfg: PathConfig, batch: int, seed: int = 0):
    """depth (B,N,D,H,W) softmax over D; context (B,N,C,H,W) ~ N(0,1). fp32, CPU."""
    H, W = cfg.feat_hw
    g = torch.Generator().manual_seed(seed)
    depth = (torch.randn(batch, cfg.n_cams, cfg.D, H, W, generator=g) * 3.0).softmax(dim=2)
    g2 = torch.Generator().manual_seed(seed + 1)
    context = torch.randn(batch, cfg.n_cams, cfg.channels, H, W, generator=g2)
    return depth.contiguous(), context.contiguous()


# ------------------------------------------------------------------------------------------------ the path's training step (round 6)
PYRAMID = lambda H, W: [(H, W), (2 * H, 2 * W), (H // 2, W // 2), (H // 4, W // 4)]     # noqa: E731  level 0 = the depth net's level


def fb_path_step(name, B, levels, dev, seed=0, feat_grad=True, train=True):
    """FBViewTransform (forward projection + backward projection + re-add, fbocc.py:344-366) at a named workload with seeded inputs,
    and `step()` = one forward + backward with the upstream gradient HANDED OVER in the output's own memory layout (what the voxel
    encoder's backward gives the path in training; bench.py `fb_projection_train`, tools/train_path.py).  The sampling_offsets /
    attention_weights heads are randomised (the reference init zeroes them: offsets / weights would not depend on the queries).
    -> dict(pc, model, cam, depth, ctx, mlvl, step, leaves, names, gout)"""
    from . import configs
    from .fb_view_transform import FBViewTransform
    pc = CONFIGS[name]
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample,
                            num_levels=levels)
    torch.manual_seed(seed)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection'])
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if 'sampling_offsets.weight' in n_ or 'attention_weights.weight' in n_:
                p_.normal_(0, 0.05)
    m = m.to(dev)
    m = m.train() if train else m.eval()
    cam = [t.to(dev) for t in camera_rig(pc, B, seed=0, bda_aug=True)]
    depth, ctx = depth_and_context(pc, B, seed=0)
    depth, ctx = depth.to(dev).requires_grad_(train), ctx.to(dev).requires_grad_(train)
    mlvl, shapes = None, [tuple(ctx.shape[-2:])]
    if levels > 1:
        H, W = ctx.shape[-2:]
        g = torch.Generator().manual_seed(5)
        shapes = PYRAMID(H, W)[:levels]
        mlvl = [torch.randn(B, pc.n_cams, pc.channels, h, w_, generator=g).to(dev).requires_grad_(train and feat_grad) for h, w_ in shapes]
        mlvl[0] = ctx
    with torch.no_grad():
        out = m(cam, ctx, depth, mlvl_feats=mlvl)
    w = torch.randn(B, pc.channels, Y, X, Z, generator=torch.Generator().manual_seed(11)).to(dev)
    gout = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=dev).copy_(w)       # the output's own layout
    del out, w
    named = list(m.named_parameters())
    leaves = [p for _, p in named] + [depth, ctx] + (list(mlvl[1:]) if mlvl else [])
    names = [n for n, _ in named] + ['depth', 'ctx'] + [f'mlvl{i}' for i in range(1, len(mlvl or []))]

    def step():
        for t in leaves:
            t.grad = None
        o = m(cam, ctx, depth, mlvl_feats=mlvl)
        o.backward(gout)
        return o
    return dict(pc=pc, cfg=cfg, gcb=gcb, model=m, cam=cam, depth=depth, ctx=ctx, mlvl=mlvl, shapes=shapes, step=step, leaves=leaves,
                names=names, gout=gout)
