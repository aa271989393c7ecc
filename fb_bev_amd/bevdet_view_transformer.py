"""BEVDet-era view transformers on the MI355X lift-splat path: `LSSViewTransformer`,
`LSSViewTransformer2`.

Reference: mmdet3d/models/necks/view_transformer.py
    :16-329    LSSViewTransformer   -- a 1x1 `depth_net` conv produces D depth logits + C context channels per pixel
                                       (:296-323), lift-splat with bev_pool_v2 (:165-192, 266-294), Z collapsed into the
                                       channel axis (`torch.cat(bev_feat.unbind(dim=2), 1)`, :191)
    :332-724   LSSViewTransformer2  -- same, but points whose depth probability is <= 0.01 are dropped before the ranking
                                       (`kept &= depth > 0.01`, :552-557), so the number of kept points P depends on the data
Geometry (:102-143) and ranking (:194-258) are the ones of the FB-OCC 3-D class, i.e. the kernels of this repo:
`fbbev_lift_rank_build` / `fbbev_rank_build_depth` for the indices, the fused dense pooling (+ its sync-free backward)
for the splat.  P stays on the device in both variants: no host synchronisation where the reference has a boolean-mask
gather, an argsort and a `where`.  Not referenced by any fb_occ config (lowest rank of SURVEY 8f); kept to the module
interface: `forward(input) -> (bev_feat, depth)`.
"""
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .view_transformer import LSSViewTransformerFunction3D


class LSSViewTransformer(LSSViewTransformerFunction3D):
    depth_threshold = None                     # LSSViewTransformer2: 0.01

    def __init__(self, grid_config, input_size, downsample=16, in_channels=512, out_channels=64, accelerate=False,
                 uniform=False, with_cp=False, **execution_knobs):
        super().__init__(grid_config, input_size, downsample=downsample, accelerate=accelerate, uniform=uniform, with_cp=with_cp,
                         **execution_knobs)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.depth_net = nn.Conv2d(in_channels, self.D + out_channels, kernel_size=1, padding=0)      # :55-56

    def _collapse_z(self, bev):
        """(B,C,Y,X,Z) view of the pooled (B,C,Z,Y,X) volume -> (B, Z*C, Y, X), channel = z*C + c  (:187-192)."""
        B, C, Y, X, Z = bev.shape
        return bev.permute(0, 4, 1, 2, 3).reshape(B, Z * C, Y, X)

    def view_transform_core(self, input, depth, tran_feat):
        """:266-289 / :645-686.  `input` = (features, rots, trans, intrins, post_rots, post_trans, bda)."""
        B, N, _, H, W = input[0].shape
        cam_params = list(input[1:7])
        depth5 = depth.view(B, N, self.D, H, W)
        feat5 = tran_feat.view(B, N, self.out_channels, H, W)
        if self.depth_threshold is None:
            bev = super().view_transform_core(cam_params, depth5, feat5)
        else:
            # data-dependent point set: indices rebuilt from the geometry AND the depth distribution (never cached)
            idx = self.build_index(self.get_lidar_coor(*cam_params), depth=depth5.detach(), depth_threshold=self.depth_threshold)
            bev = self.lift_splat(idx, depth5, feat5)
        if self.accelerate and self.depth_threshold is None:
            return bev.permute(0, 1, 4, 2, 3).squeeze(2), depth          # :287 squeezes Z (the BEVDet grids have Z = 1)
        return self._collapse_z(bev), depth

    def view_transform(self, input, depth, tran_feat):
        if self.accelerate and self.depth_threshold is None:
            self.pre_compute(list(input[1:7]))
        return self.view_transform_core(input, depth, tran_feat)

    def forward(self, input, return_depth_digit=False):
        """:296-323 -> (bev_feat (B, Z*C, Y, X), depth (B*N, D, H, W)) [+ depth logits]."""
        x = input[0]
        B, N, C, H, W = x.shape
        x = x.view(B * N, C, H, W)
        x = checkpoint(self.depth_net, x, use_reentrant=False) if (self.with_cp and x.requires_grad) else self.depth_net(x)
        depth_digit = x[:, :self.D]
        tran_feat = x[:, self.D:self.D + self.out_channels]
        depth = (depth_digit * 0 if self.uniform else depth_digit).softmax(dim=1)
        out = self.view_transform(input, depth, tran_feat)
        return out + (depth_digit,) if return_depth_digit else out

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda):
        return None


class LSSViewTransformer2(LSSViewTransformer):
    """:332-724: the depth-thresholded variant (`depth > 0.01`, :552-557)."""
    depth_threshold = 0.01
