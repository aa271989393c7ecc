"""CPU ORACLE for the FB-OCC view-transformation hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (fb_bev_amd/) never imports it and has no CPU fallback.

What is restated and from where (paths relative to /root/reference):
  * frustum template .......... forward_projection/view_transformer.py:389-411
  * get_lidar_coor ............. forward_projection/view_transformer.py:458-498
  * voxel_pooling_prepare_v2 ... forward_projection/view_transformer.py:547-605
      (incl. the fp32 rank arithmetic, trunc-toward-zero voxel index, and float compares)
  * bev_pool_v2 fwd/bwd ........ ops/bev_pool_v2/src/bev_pool_cuda.cu:18-45,64-118 (C, fbbev_oracle.c)
      + wrapper semantics ...... ops/bev_pool_v2/bev_pool.py:14-89
  * ms_deform_attn fwd/bwd ..... mmcv-full 1.5.2 (external; restated, PARITY UNPINNED) (C)
      + F.grid_sample formulation (mmcv multi_scale_deformable_attn_pytorch semantics)
  * point_sampling / reference points ... backward_projection/bevformer_utils/bevformer_encoder.py:52-120

Pinning: tests/golden/*.npz were produced by importing the REAL reference Python
(tests/golden/make_golden.py, run in the build container where /root/reference is mounted)
and tests/test_oracle_golden.py checks this restatement against them, plus the 4-point
known-answer fixture of bev_pool.py:144-175.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    """Load (building on first use) the C oracle."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, '_build', 'libfbbev_oracle.so')
        src = os.path.join(_HERE, 'fbbev_oracle.c')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['make', '-C', _HERE, '_build/libfbbev_oracle.so'],
                                  stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _i32(t):
    return t.detach().to(torch.int32).contiguous()


# --------------------------------------------------------------------------- bev_pool_v2
def bev_pool_v2_fwd(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                    interval_starts, interval_lengths, use_fma=True):
    """Loop-exact bev_pool_v2 forward. Returns out (B,Z,Y,X,C) like QuickCumsumCuda.forward
    (bev_pool.py:14-38): zero-initialised, only non-empty voxels written."""
    depth, feat = _f32(depth), _f32(feat)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    st, ln = _i32(interval_starts), _i32(interval_lengths)
    out = torch.zeros(tuple(bev_feat_shape), dtype=torch.float32)
    c = feat.shape[-1]
    lib().oracle_bev_pool_v2_fwd(c, st.numel(), _p(depth), _p(feat), _p(rd), _p(rf), _p(rb),
                                 _p(st), _p(ln), _p(out), int(use_fma))
    return out


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths, use_fma=True):
    """bev_pool.py:83-89: op output permuted to (B,C,Z,Y,X) contiguous."""
    out = bev_pool_v2_fwd(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                          interval_starts, interval_lengths, use_fma)
    return out.permute(0, 4, 1, 2, 3).contiguous()


def intervals_from_sorted(keys):
    """Interval starts/lengths over a sorted int key array (view_transformer.py:593-602,
    bev_pool.py:47-54)."""
    n = keys.shape[0]
    kept = torch.ones(n, dtype=torch.bool)
    kept[1:] = keys[1:] != keys[:-1]
    starts = torch.where(kept)[0].int()
    lengths = torch.zeros_like(starts)
    if starts.numel():
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = n - starts[-1]
    return starts, lengths


def bev_pool_v2_bwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, use_fma=True):
    """QuickCumsumCuda.backward (bev_pool.py:40-80): re-sort by ranks_feat (stable here; the
    reference's argsort is unstable, which only permutes the fp32 summation order), rebuild the
    intervals over ranks_feat, run the grad kernel. Returns (depth_grad, feat_grad)."""
    depth, feat, out_grad = _f32(depth), _f32(feat), _f32(out_grad)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    order = torch.argsort(rf, stable=True)
    rf, rd, rb = rf[order].contiguous(), rd[order].contiguous(), rb[order].contiguous()
    st, ln = intervals_from_sorted(rf)
    depth_grad = torch.zeros_like(depth)
    feat_grad = torch.zeros_like(feat)
    c = out_grad.shape[-1]
    lib().oracle_bev_pool_v2_bwd(c, st.numel(), _p(out_grad), _p(depth), _p(feat), _p(rd), _p(rf),
                                 _p(rb), _p(st.contiguous()), _p(ln.contiguous()),
                                 _p(depth_grad), _p(feat_grad), int(use_fma))
    return depth_grad, feat_grad


def bev_pool_v2_torch(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape):
    """Pure-PyTorch CPU formulation: index_add of depth*feat rows.
    This is the 'reference pure-CPU PyTorch path' timed by bench.py's cpu_baseline."""
    B, Z, Y, X, C = bev_feat_shape
    out = torch.zeros(B * Z * Y * X, C, dtype=torch.float32)
    contrib = depth.reshape(-1)[ranks_depth.long(), None] * feat.reshape(-1, C)[ranks_feat.long()]
    out.index_add_(0, ranks_bev.long(), contrib)
    return out.view(B, Z, Y, X, C).permute(0, 4, 1, 2, 3).contiguous()


# --------------------------------------------------------------------------- geometry + ranking
class ViewTransformerOracle:
    """Restates LSSViewTransformerFunction3D's index build (view_transformer.py:335-411,458-605)."""

    def __init__(self, grid_config, input_size, downsample):
        x, y, z = grid_config['x'], grid_config['y'], grid_config['z']
        # :384-387 -- python-float arithmetic, then fp32 tensors
        self.grid_lower_bound = torch.Tensor([c[0] for c in (x, y, z)])
        self.grid_interval = torch.Tensor([c[2] for c in (x, y, z)])
        self.grid_size = torch.Tensor([(c[1] - c[0]) / c[2] for c in (x, y, z)])
        H_in, W_in = input_size
        Hf, Wf = H_in // downsample, W_in // downsample
        d = torch.arange(*grid_config['depth'], dtype=torch.float).view(-1, 1, 1).expand(-1, Hf, Wf)
        self.D = d.shape[0]
        xs = torch.linspace(0, W_in - 1, Wf, dtype=torch.float).view(1, 1, Wf).expand(self.D, Hf, Wf)
        ys = torch.linspace(0, H_in - 1, Hf, dtype=torch.float).view(1, Hf, 1).expand(self.D, Hf, Wf)
        self.frustum = torch.stack((xs, ys, d), -1)  # (D,H,W,3)  :389-411

    def get_lidar_coor(self, rots, trans, cam2imgs, post_rots, post_trans, bda):
        """:458-498. image-plane frustum -> undo image aug -> unproject -> cam->ego -> bda."""
        B, N, _ = trans.shape
        pts = self.frustum.to(rots) - post_trans.view(B, N, 1, 1, 1, 3)
        pts = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
        pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
        combine = rots.matmul(torch.inverse(cam2imgs))
        pts = combine.view(B, N, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1)
        pts = pts + trans.view(B, N, 1, 1, 1, 3)
        pts = bda.view(B, 1, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1)).squeeze(-1)
        return pts

    def voxel_pooling_prepare_v2(self, coor):
        """:547-605, with the order canonicalised: the reference's argsort() is unstable, the
        oracle uses the stable order (ascending point id inside a voxel)."""
        B, N, D, H, W, _ = coor.shape
        npts = B * N * D * H * W
        ranks_depth = torch.arange(0, npts, dtype=torch.int)
        ranks_feat = torch.arange(0, npts // D, dtype=torch.int).reshape(B, N, 1, H, W)
        ranks_feat = ranks_feat.expand(B, N, D, H, W).flatten()
        vox = ((coor - self.grid_lower_bound.to(coor)) / self.grid_interval.to(coor))
        vox = vox.long().view(npts, 3)  # .long(): truncation toward zero (trap ii)
        bidx = torch.arange(0, B).reshape(B, 1).expand(B, npts // B).reshape(npts, 1).to(vox)
        vox = torch.cat((vox, bidx), 1)
        gs = self.grid_size
        kept = (vox[:, 0] >= 0) & (vox[:, 0] < gs[0]) & (vox[:, 1] >= 0) & (vox[:, 1] < gs[1]) & \
               (vox[:, 2] >= 0) & (vox[:, 2] < gs[2])
        if len(kept) == 0 or not bool(kept.any()):
            return None, None, None, None, None
        vox, ranks_depth, ranks_feat = vox[kept], ranks_depth[kept], ranks_feat[kept]
        # long * 0-dim fp32 tensor promotes to fp32: the rank is evaluated in float32 (trap i)
        ranks_bev = vox[:, 3] * (gs[2] * gs[1] * gs[0])
        ranks_bev += vox[:, 2] * (gs[1] * gs[0])
        ranks_bev += vox[:, 1] * gs[0] + vox[:, 0]
        assert ranks_bev.dtype == torch.float32
        order = ranks_bev.argsort(stable=True)
        ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
        starts, lengths = intervals_from_sorted(ranks_bev)
        return (ranks_bev.int().contiguous(), ranks_depth.int().contiguous(),
                ranks_feat.int().contiguous(), starts.int().contiguous(), lengths.int().contiguous())

    def bev_feat_shape(self, B, C):
        return (B, int(self.grid_size[2]), int(self.grid_size[1]), int(self.grid_size[0]), C)

    def view_transform(self, cam_params, depth, tran_feat, use_fma=True):
        """:521-545,613-643 -> bev_feat (B,C,Y,X,Z) view."""
        coor = self.get_lidar_coor(*cam_params)
        rb, rd, rf, st, ln = self.voxel_pooling_prepare_v2(coor)
        B, _, C = tran_feat.shape[0], None, tran_feat.shape[2]
        if rb is None:
            return torch.zeros(B, C, int(self.grid_size[0]), int(self.grid_size[1]), int(self.grid_size[2]))
        feat = tran_feat.permute(0, 1, 3, 4, 2)
        out = bev_pool_v2(depth, feat, rd, rf, rb, self.bev_feat_shape(B, C), st, ln, use_fma)
        return out.permute(0, 1, 3, 4, 2)


def canonicalize(ranks_bev, ranks_depth, ranks_feat):
    """Canonical stable order: ascending ranks_depth inside equal ranks_bev."""
    key = ranks_bev.long() * (int(ranks_depth.max()) + 1 if ranks_depth.numel() else 1) + ranks_depth.long()
    order = torch.argsort(key, stable=True)
    return ranks_bev[order], ranks_depth[order], ranks_feat[order]


# --------------------------------------------------------------------------- MSDeformAttn
def _i64(t):
    return t.detach().to(torch.int64).contiguous()


def msda_fwd(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """Loop-exact im2col (C). value (B,S,M,Dh), loc (B,Q,M,L,P,2), w (B,Q,M,L,P) -> (B,Q,M*Dh)."""
    value, loc, w = _f32(value), _f32(sampling_locations), _f32(attention_weights)
    ss, ls = _i64(spatial_shapes), _i64(level_start_index)
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    out = torch.zeros(B, Q, M * Dh, dtype=torch.float32)
    lib().oracle_msda_fwd(_p(value), _p(ss), _p(ls), _p(loc), _p(w), B, S, M, Dh, L, Q, P, _p(out))
    return out


def msda_bwd(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output):
    value, loc, w = _f32(value), _f32(sampling_locations), _f32(attention_weights)
    go = _f32(grad_output)
    ss, ls = _i64(spatial_shapes), _i64(level_start_index)
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    gv, gl, gw = torch.zeros_like(value), torch.zeros_like(loc), torch.zeros_like(w)
    lib().oracle_msda_bwd(_p(value), _p(ss), _p(ls), _p(loc), _p(w), _p(go), B, S, M, Dh, L, Q, P,
                          _p(gv), _p(gl), _p(gw))
    return gv, gl, gw


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """F.grid_sample formulation (semantics of mmcv's multi_scale_deformable_attn_pytorch, the CPU
    function the reference imports at spatial_cross_attention_depth.py:7). Differentiable."""
    import torch.nn.functional as F
    B, _, M, Dh = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes.tolist()]
    value_list = value.split(sizes, dim=1)
    grids = (2 * sampling_locations - 1).to(value.dtype)
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes.tolist()):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(B * M, Dh, int(h), int(w))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)  # (B*M, Q, P, 2)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    aw = attention_weights.to(value.dtype).transpose(1, 2).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(B, M * Dh, Q)
    return out.transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- backward projection geometry
def reference_points_3d(grid_config_bevformer):
    """bevformer_encoder.py:66-75: voxel-centre coordinates (Y, X, Za, 3)."""
    xb, yb, zb = (grid_config_bevformer[k] for k in ('x', 'y', 'z'))
    X = torch.arange(*xb, dtype=torch.float) + xb[-1] / 2
    Y = torch.arange(*yb, dtype=torch.float) + yb[-1] / 2
    Z = torch.arange(*zb, dtype=torch.float) + zb[-1] / 2
    Y, X, Z = torch.meshgrid([Y, X, Z], indexing='ij')
    return torch.stack([X, Y, Z], dim=-1)


def inv3x3_closed_form(m):
    """Adjugate/determinant inverse (what the product's point_sampling uses instead of LU)."""
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    A, Bc, C = e * i - f * h, -(d * i - f * g), d * h - e * g
    det = a * A + b * Bc + c * C
    adj = torch.stack([torch.stack([A, -(b * i - c * h), b * f - c * e], -1),
                       torch.stack([Bc, a * i - c * g, -(a * f - c * d)], -1),
                       torch.stack([C, -(a * h - b * g), a * e - b * d], -1)], -2)
    return adj / det[..., None, None]


def point_sampling(reference_points, cam_params, final_dim, inverse=torch.inverse):
    """bevformer_encoder.py:91-120: ego voxel centres -> per-camera normalised pixel coords,
    in-image mask and camera-frame depth.  `inverse` = torch.inverse reproduces the reference; tests
    that compare against the GPU module pass the closed-form inverse so both sides see the same mask."""
    rots, trans, intrins, post_rots, post_trans, bda = cam_params
    B, N, _ = trans.shape
    eps = 1e-5
    ogfH, ogfW = final_dim
    rp = reference_points.to(trans)[None, None].repeat(B, N, 1, 1, 1, 1)
    rp = inverse(bda).view(B, 1, 1, 1, 1, 3, 3).matmul(rp.unsqueeze(-1)).squeeze(-1)
    rp = rp - trans.view(B, N, 1, 1, 1, 3)
    combine = inverse(rots.matmul(inverse(intrins)))
    cam = combine.view(B, N, 1, 1, 1, 3, 3).matmul(rp.unsqueeze(-1)).squeeze(-1)
    cam = torch.cat([cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps),
                     cam[..., 2:3]], 5)
    cam = post_rots.view(B, N, 1, 1, 1, 3, 3).matmul(cam.unsqueeze(-1)).squeeze(-1)
    cam = cam + post_trans.view(B, N, 1, 1, 1, 3)
    cam = torch.cat([cam[..., 0:1] / ogfW, cam[..., 1:2] / ogfH, cam[..., 2:3]], -1)
    mask = (cam[..., 2:3] > eps) & (cam[..., 0:1] > eps) & (cam[..., 0:1] < (1.0 - eps)) & \
           (cam[..., 1:2] > eps) & (cam[..., 1:2] < (1.0 - eps))
    B, N, H, W, Dz, _ = cam.shape
    cam = cam.permute(1, 0, 2, 3, 4, 5).reshape(N, B, H * W, Dz, 3)
    mask = mask.permute(1, 0, 2, 3, 4, 5).reshape(N, B, H * W, Dz, 1).squeeze(-1)
    return cam[..., :2], mask, cam[..., 2:3]
