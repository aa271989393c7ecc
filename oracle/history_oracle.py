"""CPU oracle of FB-OCC's temporal history fusion -- TEST INFRASTRUCTURE ONLY.

Restates mmdet3d/models/fbbev/detectors/fbocc.py:
  * generate_forward_transformation_matrix   :36-41
  * FBOCC.generate_grid                       :169-205   (rt_flow + normalised sampling grid)
  * FBOCC.fuse_history                        :207-319   (history state machine, warp, time conv, cat conv)
and the 5-D trilinear `F.grid_sample(..., align_corners=True, mode='bilinear')` (zero padding) of :275 as an
explicit gather (ATen grid_sampler_3d arithmetic: unnormalise ((g+1)/2)*(size-1), floor corners, weights as
products of the opposite-corner distances, corners accumulated in the order tnw,tne,tsw,tse,bnw,bne,bsw,bse).

Pinned by tests/golden/history_fusion_seq4.npz, produced by the real reference methods
(tests/golden/make_golden_history.py).  Only tests/ may import this module.
"""
import torch
import torch.nn.functional as F


def forward_aug_matrix(bda):
    """fbocc.py:36-41: (B,3,3) -> homogeneous (B,4,4)."""
    b = bda.shape[0]
    m = torch.eye(4, dtype=bda.dtype)[None].repeat(b, 1, 1)
    m[:, :3, :3] = bda
    return m


def feat2bev_matrix(dx, bx, dtype=torch.float32):
    """fbocc.py:184-195: voxel index -> metres."""
    m = torch.zeros(4, 4, dtype=dtype)
    m[0, 0], m[1, 1], m[2, 2] = dx[0], dx[1], dx[2]
    m[0, 3] = bx[0] - dx[0] / 2.
    m[1, 3] = bx[1] - dx[1] / 2.
    m[2, 3] = bx[2] - dx[2] / 2.
    m[3, 3] = 1
    return m.view(1, 4, 4)


def rt_flow(history_forward_augs, forward_augs, curr_to_prev_ego_rt, dx, bx):
    """fbocc.py:197-203: current voxel index -> previous frame's voxel index, (B,4,4)."""
    f2b = feat2bev_matrix(dx, bx, forward_augs.dtype)
    return torch.inverse(f2b) @ history_forward_augs @ curr_to_prev_ego_rt @ torch.inverse(forward_augs) @ f2b


def generate_grid(flow, zyx, dtype=torch.float32):
    """fbocc.py:172-205 given rt_flow: normalised grid (B, Y, X, Z, 3), last dim (x, y, z)."""
    z, h, w = zyx
    n = flow.shape[0]
    xs = torch.linspace(0, w - 1, w, dtype=dtype).view(1, w, 1).expand(h, w, z)
    ys = torch.linspace(0, h - 1, h, dtype=dtype).view(h, 1, 1).expand(h, w, z)
    zs = torch.linspace(0, z - 1, z, dtype=dtype).view(1, 1, z).expand(h, w, z)
    grid = torch.stack((xs, ys, zs, torch.ones_like(xs)), -1).view(1, h, w, z, 4).expand(n, h, w, z, 4)
    grid = grid.reshape(n, h, w, z, 4, 1)
    grid = flow.view(n, 1, 1, 1, 4, 4) @ grid
    norm = torch.tensor([w - 1.0, h - 1.0, z - 1.0], dtype=dtype)
    return grid[:, :, :, :, :3, 0] / norm.view(1, 1, 1, 1, 3) * 2.0 - 1.0


def grid_sample_3d(inp, grid):
    """inp (N,C,D,H,W), grid (N,Do,Ho,Wo,3) in [-1,1], align_corners=True, zeros padding, trilinear."""
    N, C, D, H, W = inp.shape
    ix = ((grid[..., 0] + 1.) / 2.) * (W - 1)
    iy = ((grid[..., 1] + 1.) / 2.) * (H - 1)
    iz = ((grid[..., 2] + 1.) / 2.) * (D - 1)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    flat = inp.reshape(N, C, D * H * W)
    out = torch.zeros((N, C) + tuple(grid.shape[1:4]), dtype=inp.dtype)
    # corner order and weights of ATen's grid_sampler_3d (t = z0, b = z1; n = y0, s = y1; w = x0, e = x1)
    corners = [(x0, y0, z0, (x1 - ix) * (y1 - iy) * (z1 - iz)), (x1, y0, z0, (ix - x0) * (y1 - iy) * (z1 - iz)),
               (x0, y1, z0, (x1 - ix) * (iy - y0) * (z1 - iz)), (x1, y1, z0, (ix - x0) * (iy - y0) * (z1 - iz)),
               (x0, y0, z1, (x1 - ix) * (y1 - iy) * (iz - z0)), (x1, y0, z1, (ix - x0) * (y1 - iy) * (iz - z0)),
               (x0, y1, z1, (x1 - ix) * (iy - y0) * (iz - z0)), (x1, y1, z1, (ix - x0) * (iy - y0) * (iz - z0))]
    for cx, cy, cz, wgt in corners:
        ok = (cx >= 0) & (cx <= W - 1) & (cy >= 0) & (cy <= H - 1) & (cz >= 0) & (cz <= D - 1)
        lin = (cz.clamp(0, D - 1) * H + cy.clamp(0, H - 1)) * W + cx.clamp(0, W - 1)
        val = torch.gather(flat, 2, lin.long().reshape(N, 1, -1).expand(N, C, -1)).reshape(out.shape)
        out = out + val * (wgt * ok.to(inp.dtype))[:, None]
    return out


def warp_history(history, flow):
    """history (B,CH,Z,Y,X), flow (B,4,4) -> sampled (B,CH,Z,Y,X) (fbocc.py:267-275)."""
    Z, Y, X = history.shape[2:]
    grid = generate_grid(flow, (Z, Y, X), history.dtype)
    return grid_sample_3d(history, grid.permute(0, 3, 1, 2, 4))


class HistoryFusionOracle:
    """fuse_history with explicit weights (conv + eval-mode batch norm + ReLU as plain tensor algebra).

    weights: dict with time_w (C,C+1), time_b (C), time_bn = (weight, bias, mean, var, eps) and the same for cat_*.
    """

    def __init__(self, weights, dx, bx, history_cat_num, channels, sweep_freq=0.5, do_history=True):
        self.w, self.dx, self.bx = weights, dx, bx
        self.T, self.C, self.freq, self.do_history = history_cat_num, channels, sweep_freq, do_history
        self.history_bev = None
        self.history_seq_ids = self.history_forward_augs = self.history_sweep_time = None

    @staticmethod
    def _conv_bn_relu(x, w, b, bn):
        """x (N,Cin,Z,Y,X): 1x1x1 conv + eval BatchNorm + ReLU"""
        g, beta, mean, var, eps = bn
        y = torch.einsum('oc,nczyx->nozyx', w, x) + b.view(1, -1, 1, 1, 1)
        y = (y - mean.view(1, -1, 1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1, 1) + eps) * g.view(1, -1, 1, 1, 1) + \
            beta.view(1, -1, 1, 1, 1)
        return y.clamp_min(0)

    def fuse(self, curr_bev, seq_ids, start_of_sequence, curr_to_prev_ego_rt, bda):
        """curr_bev (B,C,Y,X,Z) -> (B,Cout,Y,X,Z); also returns the sampled history (for the warp tests)."""
        T, C = self.T, self.C
        curr = curr_bev.permute(0, 1, 4, 2, 3)                                   # :212 n,c,z,h,w
        fwd = forward_aug_matrix(bda)                                            # :220
        if self.history_bev is None:                                             # :227-238
            self.history_bev = curr.repeat(1, T, 1, 1, 1)
            self.history_seq_ids = seq_ids.clone()
            self.history_forward_augs = fwd.clone()
            self.history_sweep_time = curr.new_zeros(curr.shape[0], T)
        assert int((self.history_seq_ids != seq_ids)[~start_of_sequence].sum()) == 0   # :248
        self.history_sweep_time = self.history_sweep_time + 1                    # :252
        if start_of_sequence.any():                                              # :253-261
            self.history_bev[start_of_sequence] = curr[start_of_sequence].repeat(1, T, 1, 1, 1)
            self.history_sweep_time[start_of_sequence] = 0
            self.history_seq_ids[start_of_sequence] = seq_ids[start_of_sequence]
            self.history_forward_augs[start_of_sequence] = fwd[start_of_sequence]
        flow = rt_flow(self.history_forward_augs, fwd, curr_to_prev_ego_rt, self.dx, self.bx)
        sampled = warp_history(self.history_bev, flow)                           # :267-275
        sweep = torch.cat([self.history_sweep_time.new_zeros(curr.shape[0], 1), self.history_sweep_time], 1)  # :279-281
        feats_cat = torch.cat([curr, sampled], 1)                                # :286
        B, _, Z, Y, X = feats_cat.shape
        f = feats_cat.reshape(B, T + 1, C, Z, Y, X)                              # :289-290
        tchan = (sweep * self.freq)[:, :, None, None, None, None].expand(B, T + 1, 1, Z, Y, X)
        f = torch.cat([f, tchan], 2)                                             # :292-295
        y = self._conv_bn_relu(f.reshape(-1, C + 1, Z, Y, X), self.w['time_w'], self.w['time_b'], self.w['time_bn'])
        y = y.reshape(B, (T + 1) * C, Z, Y, X)                                   # :303-310
        out = self._conv_bn_relu(y, self.w['cat_w'], self.w['cat_b'], self.w['cat_bn'])
        self.history_bev = feats_cat[:, :-C].clone()                             # :312
        self.history_sweep_time = sweep[:, :-1]                                  # :313
        self.history_forward_augs = fwd.clone()                                  # :314
        if not self.do_history:                                                  # :317-318
            self.history_bev = None
        return out.permute(0, 1, 3, 4, 2).clone(), sampled, flow


def weights_from_state_dict(sd, eps=1e-5):
    """state_dict of FBOCC's two Sequentials (fbocc.py:111-127) -> the oracle's weight dict."""
    t = lambda k: torch.as_tensor(sd[k])  # noqa: E731
    return {
        'time_w': t('history_keyframe_time_conv.0.weight').flatten(1), 'time_b': t('history_keyframe_time_conv.0.bias'),
        'time_bn': (t('history_keyframe_time_conv.1.weight'), t('history_keyframe_time_conv.1.bias'),
                    t('history_keyframe_time_conv.1.running_mean'), t('history_keyframe_time_conv.1.running_var'), eps),
        'cat_w': t('history_keyframe_cat_conv.0.weight').flatten(1), 'cat_b': t('history_keyframe_cat_conv.0.bias'),
        'cat_bn': (t('history_keyframe_cat_conv.1.weight'), t('history_keyframe_cat_conv.1.bias'),
                   t('history_keyframe_cat_conv.1.running_mean'), t('history_keyframe_cat_conv.1.running_var'), eps),
    }


def grid_sample_reference(inp, grid):
    """torch's own kernel, used only to cross-check grid_sample_3d above."""
    return F.grid_sample(inp, grid, align_corners=True, mode='bilinear')
