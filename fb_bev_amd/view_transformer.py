"""Forward projection (lift-splat) -- host-side mirror of
mmdet3d/models/fbbev/view_transformation/forward_projection/view_transformer.py:315-663
(`LSSViewTransformerFunction3D`): same constructor arguments, attribute names and method names,
so an `occupancy_configs/fb_occ/*.py` `forward_projection=dict(type='LSSViewTransformerFunction3D', ...)`
entry builds this class unchanged.

Two execution paths over the same HIP kernels:
  * reference-shaped : get_lidar_coor -> voxel_pooling_prepare_v2 (exact-size index tensors, one
    host sync to read the two counts) -> bev_pool_v2 (autograd op)           [parity surface]
  * fused (default)  : fbbev_lift_rank_build (geometry evaluated inside the sort's first pass, no coor
    tensor) -> fbbev_pool_tile_index -> fbbev_bev_pool_v2_dense_fwd, all enqueued on the current stream
    with device-side counts: no host sync, no new_zeros, no permute().contiguous()   [bench surface]
Both produce the same bits: the pooled sums are the same in-order fmaf chains.
"""
import torch
import torch.nn as nn

from . import _capi
from .bev_pool import bev_pool_v2

__all__ = ['LSSViewTransformerFunction3D', 'LiftSplat', 'gen_dx_bx']


def gen_dx_bx(xbound, ybound, zbound):
    """view_transformer.py:17-21."""
    rows = (xbound, ybound, zbound)
    dx = torch.Tensor([r[2] for r in rows])
    bx = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
    nx = torch.Tensor([(r[1] - r[0]) / r[2] for r in rows])
    return dx, bx, nx


class _IndexSet:
    """Device-resident index tensors of one rank build, padded to the frustum size n; the valid
    prefixes are counts[0]=P points and counts[1]=I intervals (device-side)."""
    __slots__ = ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths',
                 'interval_rank', 'counts', 'n', 'cache_state', 'tile_tables')

    def __init__(self, n, device):
        def buf():
            return torch.empty(n, dtype=torch.int32, device=device)
        self.n = n
        self.ranks_bev, self.ranks_depth, self.ranks_feat = buf(), buf(), buf()
        self.interval_starts, self.interval_lengths, self.interval_rank = buf(), buf(), buf()
        self.counts = torch.empty(2, dtype=torch.int32, device=device)   # written by the rank build itself
        self.cache_state = None             # device flag of the camera-keyed cache (None: built unconditionally)
        # camera-keyed cache only: tile_voxels -> (tile table, gate).  The tables BELONG to this index set (never shared
        # with the per-call path or another cache entry) and each carries the build number it was built for, so a
        # cache hit keeps a table only if that very table was built for the current index set.
        self.tile_tables = None

    def exact(self):
        """Trim to exact sizes (ONE host sync: reads the two counts)."""
        P, I = self.counts.tolist()
        return (self.ranks_bev[:P], self.ranks_depth[:P], self.ranks_feat[:P],
                self.interval_starts[:I], self.interval_lengths[:I])


class _WorkspaceCache:
    """Per-module, per-device scratch of the fused backward (one live buffer per device, grown on demand).  Held by the
    module -- not by the process: two view transformers, or two devices, never share or free each other's workspace."""

    def __init__(self):
        self._bwd = {}

    def bwd_workspace(self, device, nbytes):
        buf = self._bwd.get(device)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)   # the old one is released by the caching
            self._bwd[device] = buf                                        # allocator in stream order
        return buf


class LiftSplat(torch.autograd.Function):
    """Fused dense pooling: depth (B,N,D,H,W), feat (B,N,H,W,C) -> (B,C,Z,Y,X), every voxel written
    once.  Backward regroups by feature pixel and runs the wave-per-pixel grad kernel."""

    @staticmethod
    def forward(ctx, depth, feat, idx, grid_zyx, tile_ws, tile_voxels, pool_flags, out_dtype=torch.float32,
                ws_cache=None, cache_state=None, table_gate=None, with_zmean=False):
        depth = depth.contiguous().float()
        # `feat` arrives as the (B,N,H,W,C) permuted view of the NCHW context (view_transformer.py:536); when
        # the underlying tensor is contiguous NCHW the copy is done by the LDS-tiled transpose kernel
        base = feat.permute(0, 1, 4, 2, 3)
        if base.is_contiguous() and base.dtype == torch.float32:
            feat = _capi.nchw_to_nhwc(base)
        else:
            feat = feat.contiguous().float()
        B, C = depth.shape[0], feat.shape[-1]
        Z, Y, X = grid_zyx
        out = torch.empty((B, C, Z, Y, X), dtype=out_dtype, device=depth.device)   # 16-bit: fp32 sums rounded at the store
        _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X,
                              tile_ws, tile_voxels, cache_state=cache_state, table_gate=table_gate)
        _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,
                                    idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, out,
                                    tile_ws, tile_voxels, pool_flags)
        ctx.idx = idx
        ctx.ws_cache = ws_cache if ws_cache is not None else _WorkspaceCache()
        ctx.grid_zyx = (Z, Y, X)
        ctx.save_for_backward(depth, feat)
        ctx.set_materialize_grads(False)                               # an unused output's gradient arrives as None, not as zeros
        if with_zmean:
            # second output: the volume's Z-mean (fbocc.py:359), one HBM-bound pass over the fresh volume; its gradient comes
            # back as `zmean_grad` and is folded into the pooling backward's read of out_grad (no expand + add over the volume)
            ctx.with_zmean = True
            return out, _capi.volume_zreduce(out.permute(0, 1, 3, 4, 2), Z)
        ctx.with_zmean = False
        return out

    @staticmethod
    def backward(ctx, out_grad, zmean_grad=None):
        """Sync-free: the frustum structure is the feature-pixel index (no argsort of ranks_feat, no mask-built
        intervals, no permuted copy of the gradient) -- fbbev_bev_pool_v2_dense_bwd."""
        depth, feat = ctx.saved_tensors
        idx = ctx.idx
        Z, Y, X = ctx.grid_zyx
        B, N, D, H, W = depth.shape
        C = feat.shape[-1]
        og = out_grad
        if og is None and (not ctx.with_zmean or zmean_grad is None):
            return (None,) * 12
        if og is None:                                                # only the Z-mean was used downstream
            og = torch.zeros((B, C, Z, Y, X), dtype=torch.float32, device=depth.device)
        sb, sc = og.stride(0), og.stride(1)
        if (og.dtype != torch.float32 or og.stride()[2:] != (Y * X, X, 1) or sc < Z * Y * X or sb < C * sc or sc % 4
                or sb % 4 or og.data_ptr() % 16):
            zl = og.permute(0, 1, 3, 4, 2)                        # the module hands the volume out as a (B,C,Y,X,Z) view
            if _capi.volume_zlast_supported(zl):                  # a gradient contiguous in THAT shape: one coalesced re-layout
                og = _capi.volume_z_to_front(zl)
            else:
                og = og.contiguous().float()
        ws = ctx.ws_cache.bwd_workspace(depth.device, _capi.pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X))
        depth_grad, feat_grad = torch.empty_like(depth), torch.empty_like(feat)
        zg = None
        if ctx.with_zmean and zmean_grad is not None:
            zg = zmean_grad.contiguous().float()
        _capi.bev_pool_v2_dense_bwd(og, depth, feat, idx.ranks_depth, idx.interval_rank, idx.interval_starts,
                                    idx.counts, idx.n, (Z, Y, X), depth_grad, feat_grad, ws, zgrad=zg, zscale=1.0 / Z)
        return depth_grad, feat_grad, None, None, None, None, None, None, None, None, None, None



import os as _os
POOL_READ_AHEAD = _os.environ.get('FBBEV_POOL_READ_AHEAD', '1') != '0'   # A/B knob: read the final pooling's gather sources once before it
POOL_READ_AHEAD_MIN_QUERIES = int(_os.environ.get('FBBEV_POOL_READ_AHEAD_MIN_QUERIES', '120000'))
_ZMEAN_CSPLIT = int(_os.environ.get('FBBEV_ZMEAN_CSPLIT', '0'))      # tuning knob: channel groups of the Z-mean kernel
_ZMEAN_ZGROUPS = int(_os.environ.get('FBBEV_ZMEAN_ZGROUPS', '0'))    # tuning knob: workgroups the Z planes of a tile are dealt to (0 = heuristic)


class LSSViewTransformerFunction3D(nn.Module):
    """Lift-Splat view transformer with a 3-D (X,Y,Z) voxel grid -- view_transformer.py:315-663.

    Args mirror the reference (grid_config, input_size, downsample, accelerate, uniform, with_cp,
    extra_relu).  `accelerate=True` keeps ONE index set per (device, B, N) keyed on the camera tensors: every call
    compares them with the cached key on the device (no host sync) and skips the rank build when nothing changed,
    rebuilds when anything did -- the `pre_compute` the reference's 3-D class intends (:607-611) but disables with
    `assert False` (:628) because nothing invalidates it.  Extra knobs: `fused` (default True) and `tile_voxels`.
    """

    def __init__(self, grid_config, input_size, downsample=16, accelerate=False, uniform=False,
                 with_cp=False, extra_relu=False, fused=True, tile_voxels=None, pool_flags=None,
                 out_dtype=torch.float32, pool_tolerance=False):
        super().__init__()
        self.uniform = uniform
        self.with_cp = with_cp
        self.extra_relu = extra_relu
        self.grid_config = grid_config
        dx, bx, nx = gen_dx_bx(grid_config['x'], grid_config['y'], grid_config['z'])
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        self.downsample = downsample
        self.create_grid_infos(**grid_config)
        self.input_size = input_size
        self.create_frustum(grid_config['depth'], input_size, downsample)
        self.accelerate = accelerate
        self.initial_flag = True
        self.fused = fused
        # storage type of the fused path's BEV volume (fp32 sums, rounded once at the store); BASELINE configs[1]
        # names bf16, configs[4] fp16; the reference itself is fp32 (bev_pool.py:16-22)
        self.out_dtype = out_dtype
        # opt-in tolerance mode of the fused fp32 pooling (FBBEV_POOL_SPLIT_LONG): intervals of more than 32 points are summed by
        # all lane groups of their workgroup -- deterministic, equal to the reference's serial chain to <= 1e-4 (north_star's
        # bar for pooled features) instead of bit for bit.  Default False: bit-exact.
        self.pool_tolerance = bool(pool_tolerance)
        self._tile_voxels_arg, self._pool_flags_arg = tile_voxels, pool_flags
        self._tiling = {}                   # number of cameras -> (tile_voxels, pool_flags)
        self.n_cams = None                  # set by the first call (the camera tensors carry it)
        self._ws = _WorkspaceCache()
        self._cache = {}
        self._index_cache = {}              # (device, B, N) -> (_IndexSet, cam_key, cache_state): accelerate=True

    # ------------------------------------------------------------------ static geometry (init time)
    def create_grid_infos(self, x, y, z, **kwargs):
        """view_transformer.py:370-387: python-float arithmetic, stored as fp32 tensors."""
        self.grid_lower_bound = torch.Tensor([cfg[0] for cfg in [x, y, z]])
        self.grid_interval = torch.Tensor([cfg[2] for cfg in [x, y, z]])
        self.grid_size = torch.Tensor([(cfg[1] - cfg[0]) / cfg[2] for cfg in [x, y, z]])

    def create_frustum(self, depth_cfg, input_size, downsample):
        """view_transformer.py:389-411: (D,H,W,3) template of (u, v, depth)."""
        H_in, W_in = input_size
        H_feat, W_feat = H_in // downsample, W_in // downsample
        self._ds = torch.arange(*depth_cfg, dtype=torch.float)
        self.D = self._ds.shape[0]
        self._xs = torch.linspace(0, W_in - 1, W_feat, dtype=torch.float)
        self._ys = torch.linspace(0, H_in - 1, H_feat, dtype=torch.float)
        d = self._ds.view(-1, 1, 1).expand(-1, H_feat, W_feat)
        x = self._xs.view(1, 1, W_feat).expand(self.D, H_feat, W_feat)
        y = self._ys.view(1, H_feat, 1).expand(self.D, H_feat, W_feat)
        self.frustum = torch.stack((x, y, d), -1)

    def tiling(self, n_cams=None):
        """(tile_voxels, pool_flags) of the dense kernel, chosen by the frustum-points-per-voxel density of THIS rig:
        measured per launch on MI355X (profiles/r01_sweep_*.jsonl).  Sparse grids (BL2: 0.4 frustum points per voxel) are
        store-bound -> 128-voxel tiles, channel range split over 2 workgroups; dense grids (shipped config: 4.2 points
        per voxel) are bound by the per-voxel gather chains -> 64-voxel tiles, no channel split."""
        n_cams = n_cams if n_cams is not None else self.n_cams
        if n_cams is None:
            raise RuntimeError('tiling() needs the number of cameras (known after the first call, or pass n_cams)')
        if n_cams not in self._tiling:
            Z, Y, X = self.grid_zyx
            density = n_cams * self.frustum.shape[0] * self.frustum.shape[1] * self.frustum.shape[2] / float(X * Y * Z)
            dense = density >= 1.0
            tv = self._tile_voxels_arg if self._tile_voxels_arg is not None else (64 if dense else _capi.DEFAULT_TILE_VOXELS)
            half = self.out_dtype != torch.float32
            tv32 = tv           # the fp32-LDS tile of this rig (write-once route: its epilogue add keeps an fp32 tile)
            if self._tile_voxels_arg is None and half and dense:
                tv *= 2     # 16-bit storage: the kernel's LDS tile is 16-bit too -> twice the voxels per workgroup at the same footprint
            # sparse grid + 16-bit storage: the SAME 20 KB of LDS hold 128 voxels x ALL channels -- one workgroup per tile, no
            # second copy of the metadata / gather chain (profiles/r02_sweep_pool_bf16_BL2_B16.jsonl: 0.321 ms vs 0.341 ms for
            # 256 voxels x 2 channel groups), chunks of 32 tiles per XCD
            fl = self._pool_flags_arg if self._pool_flags_arg is not None else (
                _capi.pool_flags(csplit=1) if dense else (_capi.pool_flags(csplit=1, swz_log2=5) if half else _capi.DEFAULT_POOL_FLAGS))
            self._tiling[n_cams] = (tv, fl, tv32)
        tv, fl = self._tiling[n_cams][:2]
        if self.pool_tolerance and self.out_dtype == torch.float32 and tv in (64, 128):
            fl |= _capi.POOL_SPLIT_LONG
        return tv, fl

    @property
    def tile_voxels(self):
        return self.tiling()[0]

    @property
    def pool_flags(self):
        return self.tiling()[1]

    @property
    def grid_zyx(self):
        return int(self.grid_size[2]), int(self.grid_size[1]), int(self.grid_size[0])

    def _axes(self, device):
        key = ('axes', device)
        if key not in self._cache:
            self._cache[key] = tuple(t.to(device).contiguous() for t in (self._xs, self._ys, self._ds))
        return self._cache[key]

    def _frustum_dev(self, device):
        key = ('frustum', device)
        if key not in self._cache:
            self._cache[key] = self.frustum.to(device).contiguous()
        return self._cache[key]

    # ------------------------------------------------------------------ per-forward geometry
    def get_lidar_coor(self, rots, trans, cam2imgs, post_rots, post_trans, bda):
        """view_transformer.py:458-498 -> coor (B,N,D,H,W,3), one HIP kernel (fbbev_lidar_coor)."""
        B, N, _ = trans.shape
        xs, ys, ds = self._axes(trans.device)
        coor = torch.empty((B, N, ds.numel(), ys.numel(), xs.numel(), 3), dtype=torch.float32,
                           device=trans.device)
        f = lambda t: t.contiguous().float()  # noqa: E731
        _capi.lidar_coor(xs, ys, ds, f(rots), f(trans), f(cam2imgs), f(post_rots), f(post_trans), f(bda), coor)
        return coor

    def _grid3(self):
        return self.grid_lower_bound.tolist(), self.grid_interval.tolist(), self.grid_size.tolist()

    def build_index(self, coor, depth=None, depth_threshold=0.01):
        """Device-side rank build (fbbev_rank_build); no host sync. -> _IndexSet
        depth: optional (B,N,D,H,W) distribution for the BEVDet-era filter `kept &= depth > 0.01`
        (mmdet3d/models/necks/view_transformer.py:556-557); P is then data dependent, still without a host sync."""
        B, N, D, H, W, _ = coor.shape
        self.n_cams = N
        n = B * N * D * H * W
        key = ('rank_ws', coor.device, n)
        if key not in self._cache:
            self._cache[key] = torch.empty(_capi.rank_workspace_bytes(n), dtype=torch.uint8, device=coor.device)
        idx = _IndexSet(n, coor.device)
        lo, it, gs = self._grid3()
        _capi.rank_build(coor.contiguous(), lo, it, gs, idx.ranks_bev, idx.ranks_depth, idx.ranks_feat,
                         idx.interval_starts, idx.interval_lengths, idx.interval_rank, idx.counts,
                         self._cache[key], depth=None if depth is None else depth.contiguous().float(),
                         depth_threshold=depth_threshold)
        return idx

    def build_index_from_cams(self, rots, trans, cam2imgs, post_rots, post_trans, bda, cached=False):
        """get_lidar_coor + voxel_pooling_prepare_v2 fused (fbbev_lift_rank_build): the keys are evaluated from
        the camera parameters inside the sort's first pass; same index tensors as build_index(get_lidar_coor()).
        cached=True (accelerate): ONE persistent index set per (device, B, N) keyed on the camera tensors -- the device
        compares them with the cached key and skips the whole build when nothing changed (no host sync), rebuilds
        when anything did (view_transformer.py:607-611 `pre_compute`, with the invalidation upstream lacks)."""
        B, N, _ = trans.shape
        self.n_cams = N
        xs, ys, ds = self._axes(trans.device)
        n = B * N * ds.numel() * ys.numel() * xs.numel()
        key = ('rank_ws', trans.device, n)
        if key not in self._cache:
            self._cache[key] = torch.empty(_capi.rank_workspace_bytes(n), dtype=torch.uint8, device=trans.device)
        cam_key = cache_state = None
        if cached:
            ck = (trans.device, B, N)
            if ck not in self._index_cache:
                self._index_cache[ck] = (_IndexSet(n, trans.device),
                                         torch.full((_capi.cam_key_words(B, N),), -1, dtype=torch.int32, device=trans.device),
                                         torch.zeros(2, dtype=torch.int32, device=trans.device))
            idx, cam_key, cache_state = self._index_cache[ck]
            if idx.tile_tables is None:
                idx.tile_tables = {}
        else:
            idx = _IndexSet(n, trans.device)
        lo, it, gs = self._grid3()
        f = lambda t: t.contiguous().float()  # noqa: E731
        _capi.lift_rank_build(xs, ys, ds, f(rots), f(trans), f(cam2imgs), f(post_rots), f(post_trans), f(bda), lo, it,
                              gs, idx.ranks_bev, idx.ranks_depth, idx.ranks_feat, idx.interval_starts,
                              idx.interval_lengths, idx.interval_rank, idx.counts, self._cache[key],
                              frustum=self._frustum_dev(trans.device), cam_key=cam_key, cache_state=cache_state)
        idx.cache_state = cache_state
        return idx

    def index_builds(self, device=None):
        """Number of real builds of the camera-keyed cache so far (ONE host sync; diagnostics / tests)."""
        return sum(int(st[1]) for (dev, _, _), (_, _, st) in self._index_cache.items() if device is None or dev == device)

    def voxel_pooling_prepare_v2(self, coor):
        """view_transformer.py:547-605 -> (ranks_bev, ranks_depth, ranks_feat, interval_starts,
        interval_lengths), exact sizes, int32; None x5 when no point falls inside the grid."""
        idx = self.build_index(coor)
        rb, rd, rf, st, ln = idx.exact()
        if st.numel() == 0:
            return None, None, None, None, None
        return rb.contiguous(), rd.contiguous(), rf.contiguous(), st.contiguous(), ln.contiguous()

    def init_acceleration_v2(self, coor):
        """view_transformer.py:500-519: exact-size index tensors as attributes (reference API; the fused path keeps its
        own camera-keyed device cache instead, see build_index_from_cams)."""
        rb, rd, rf, st, ln = self.build_index(coor).exact()
        self.ranks_bev, self.ranks_depth, self.ranks_feat = rb, rd, rf
        self.interval_starts, self.interval_lengths = st, ln

    def pre_compute(self, cam_params):
        """view_transformer.py:607-611."""
        if self.initial_flag:
            self.init_acceleration_v2(self.get_lidar_coor(*cam_params))
            self.initial_flag = False

    def _index_for(self, cam_params, *tensors):
        """The index set of this call.  accelerate=True: the camera-keyed device cache -- but only when no autograd graph
        will hold on to the index buffers (a later rebuild would overwrite what an earlier graph's backward reads)."""
        cached = self.accelerate and not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))
        return self.build_index_from_cams(*cam_params, cached=cached)

    # ------------------------------------------------------------------ pooling
    def voxel_pooling_v2(self, coor, depth, feat):
        """view_transformer.py:521-545 (reference-shaped path). feat (B,N,C,H,W) -> (B,C,X?..) view
        (B,C,Y,X,Z) exactly like the reference's final permute."""
        rb, rd, rf, st, ln = self.voxel_pooling_prepare_v2(coor)
        Z, Y, X = self.grid_zyx
        if rf is None:
            print('warning ---> no points within the predefined bev receptive field')
            return torch.zeros(size=[feat.shape[0], feat.shape[2], X, Y, Z]).to(feat)
        feat = feat.permute(0, 1, 3, 4, 2)
        bev_feat_shape = (depth.shape[0], Z, Y, X, feat.shape[-1])
        bev_feat = bev_pool_v2(depth, feat, rd, rf, rb, bev_feat_shape, st, ln)
        return bev_feat.permute(0, 1, 3, 4, 2)

    def _tile_ws(self, device, B, tile_voxels=None):
        """Tile table of the per-call path (rebuilt by every call, never trusted across calls)."""
        Z, Y, X = self.grid_zyx
        key = ('tile_ws', device, B, tile_voxels)
        if key not in self._cache:
            self._cache[key] = torch.empty(_capi.pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8,
                                           device=device)
        return self._cache[key]

    def _tile_table(self, idx, device, B, tile_voxels):
        """(tile table, gate) for this index set: a cached set owns one table per tile size, each with its own
        'built for build number' gate (-1 = never built) -- a table of another tile size, or one the per-call path
        wrote, can never be mistaken for it; a per-call set uses the module scratch and gate None (always rebuilt)."""
        if idx.cache_state is None:
            return self._tile_ws(device, B, tile_voxels), None
        if tile_voxels not in idx.tile_tables:
            Z, Y, X = self.grid_zyx
            idx.tile_tables[tile_voxels] = (
                torch.empty(_capi.pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8, device=device),
                torch.full((2,), -1, dtype=torch.int32, device=device))
        return idx.tile_tables[tile_voxels]

    def lift_splat(self, idx, depth, tran_feat, with_zmean=False):
        """Fused dense pooling on a prepared index set -> (B,C,Y,X,Z) view of (B,C,Z,Y,X) [, its Z-mean (B,C,Y,X)]."""
        feat = tran_feat.permute(0, 1, 3, 4, 2)
        table, gate = self._tile_table(idx, depth.device, depth.shape[0], self.tile_voxels)
        out = LiftSplat.apply(depth, feat, idx, self.grid_zyx, table, self.tile_voxels, self.pool_flags, self.out_dtype,
                              self._ws, idx.cache_state, gate, with_zmean)
        if with_zmean:
            return out[0].permute(0, 1, 3, 4, 2), out[1]
        return out.permute(0, 1, 3, 4, 2)

    def forward_with_zmean(self, cam_params, context, depth):
        """(volume view (B,C,Y,X,Z), its Z-mean (B,C,Y,X)) as ONE differentiable op, or None when the fused fp32 route does
        not apply (the caller then takes forward() and `.mean(-1)`)."""
        Z, Y, X = self.grid_zyx
        if (not self.fused or self.extra_relu or self.out_dtype != torch.float32 or not context.is_cuda or
                not self._fused_supported(context.shape[2]) or (Y * X) % 4 != 0):
            return None
        return self.lift_splat(self._index_for(cam_params, depth, context), depth, context, with_zmean=True)

    def view_transform_core(self, cam_params, depth, tran_feat):
        """view_transformer.py:613-635."""
        if not self.fused or not self._fused_supported(tran_feat.shape[2]):
            # reference-shaped path; also the fallback for shapes outside the fused kernels' preconditions
            # (fbbev_bev_pool_v2_dense_fwd: C % 4 == 0, C <= 256, (Y*X) % 4 == 0 -- % 8 for 16-bit storage)
            out = self.voxel_pooling_v2(self.get_lidar_coor(*cam_params), depth, tran_feat)
            return out if self.out_dtype == torch.float32 else out.to(self.out_dtype)
        return self.lift_splat(self._index_for(cam_params, depth, tran_feat), depth, tran_feat)

    def _fused_supported(self, C):
        Z, Y, X = self.grid_zyx
        q = 4 if self.out_dtype == torch.float32 else 8
        return C % 4 == 0 and C <= 256 and (Y * X) % q == 0

    def view_transform(self, cam_params, depth, tran_feat):
        """view_transformer.py:639-643 (the `pre_compute` of :641 is the camera-keyed cache inside _index_for)."""
        return self.view_transform_core(cam_params, depth, tran_feat)

    def forward(self, cam_params, context, depth, **kwargs):
        """view_transformer.py:646-660: (cam_params, context (B,N,C,H,W), depth (B,N,D,H,W)) ->
        BEV volume (B,C,Y,X,Z)."""
        bev = self.view_transform(cam_params, depth, context)
        return bev.relu() if self.extra_relu else bev

    # ------------------------------------------------------------------ write-once volume (inference, FBViewTransform)
    def pooling_inputs(self, cam_params, context, depth):
        """Index set (camera-keyed cache when accelerate=True) + the two gather sources of the fused kernels."""
        idx = self._index_for(cam_params, depth, context)
        depth = depth.contiguous().float()
        feat = _capi.nchw_to_nhwc(context.contiguous().float())
        B = depth.shape[0]
        Z, Y, X = self.grid_zyx
        tile_ws, gate = self._tile_table(idx, depth.device, B, self._wo_tile)
        _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, tile_ws,
                              self._wo_tile, cache_state=idx.cache_state, table_gate=gate)
        return idx, depth, feat, tile_ws

    @property
    def _wo_tile(self):
        # fbbev_pool_zmean takes tiles of 64..256 voxels; the write-once route adds the refined BEV in the store epilogue,
        # which keeps an fp32 LDS tile: no doubling for 16-bit storage there
        # -> the un-doubled tile of tiling(), clamped into fbbev_pool_zmean's documented 64..256 range
        self.tiling()
        return max(64, min(self._tiling[self.n_cams][2], 256))

    def pooled_zmean(self, parts):
        """bev_feat.mean(-1) of the lift-splat output, (B,C,Y,X), without materialising the volume (fbbev_pool_zmean)."""
        idx, depth, feat, tile_ws = parts
        B, C = depth.shape[0], feat.shape[-1]
        Z, Y, X = self.grid_zyx
        out = torch.empty((B, C, Y, X), dtype=torch.float32, device=depth.device)
        flags = self.pool_flags
        if _ZMEAN_CSPLIT:
            flags = (flags & ~0xF0) | ((_ZMEAN_CSPLIT & 0xF) << 4)
        self._C_hint = C
        zg = _ZMEAN_ZGROUPS if _ZMEAN_ZGROUPS else self._zmean_z_groups(B, Z, Y * X)
        partial = None
        if zg > 1:
            key = (zg, out.numel(), depth.device)
            if getattr(self, '_zmean_partial_key', None) != key:
                self._zmean_partial = torch.empty(zg * out.numel(), dtype=torch.float32, device=depth.device)
                self._zmean_partial_key = key
            partial = self._zmean_partial
        return _capi.pool_zmean(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                idx.interval_lengths, B, C, Z, Y, X, out, tile_ws, self._wo_tile, flags, z_groups=zg, partial=partial)

    def pooled_zmean_rows(self, parts, row_bias):
        """The same mean as the backward projection's query rows (B, Y*X, C) + row_bias (Y*X, C) (fbbev_pool_zmean_rows), or None when the
        launch would be split into Z groups (few tiles: pooled_zmean + the transposing pass then)."""
        idx, depth, feat, tile_ws = parts
        B, C = depth.shape[0], feat.shape[-1]
        Z, Y, X = self.grid_zyx
        self._C_hint = C
        if (_ZMEAN_ZGROUPS if _ZMEAN_ZGROUPS else self._zmean_z_groups(B, Z, Y * X)) != 1:
            return None
        flags = self.pool_flags
        if _ZMEAN_CSPLIT:
            flags = (flags & ~0xF0) | ((_ZMEAN_CSPLIT & 0xF) << 4)
        out = torch.empty((B, Y * X, C), dtype=torch.float32, device=depth.device)
        return _capi.pool_zmean_rows(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                     idx.interval_lengths, B, C, Z, Y, X, out, tile_ws, self._wo_tile, flags, row_bias=row_bias)

    def _zmean_z_groups(self, B, Z, YX):
        """fbbev_pool_zmean walks the Z planes of a tile one after the other (two barriers + two dependent round trips per plane): with
        few tiles -- the shipped grid at B = 1 has 157 -- the launch is one Z-plane latency chain per CU (106 us, the largest kernel
        of the shipped-shape S3).  Then every plane gets its own workgroup and a small reduce kernel adds the partial sums; with
        many tiles (BASELINE configs[2]: 1 252 at B = 4) the partial buffer would cost more than the chains."""
        # measured (profiles/r04_zmean_zsplit.jsonl, r04_zmean_zgroups.jsonl): shipped grid B = 1: S3 0.416 -> 0.366 ms, B = 4: 0.567 ->
        # 0.543 ms with one workgroup per plane; BASELINE configs[2] grid B = 1 (313 tiles, 205 MB of partial sums): 0.568 -> 0.64 ms,
        # shipped grid B = 16 (2 512 tiles): 1.38 -> 1.47 ms -- so: few tiles AND a partial buffer of at most 128 MB
        tiles = B * ((YX + self._wo_tile - 1) // self._wo_tile)
        partial_bytes = Z * B * self._C_hint * YX * 4 if getattr(self, '_C_hint', None) else 0
        # fbbev_pool_zmean_split takes at most 64 Z groups (include/fbbev.h): taller grids keep the single pass
        return Z if (tiles <= 1024 and partial_bytes <= (128 << 20) and Z <= 64) else 1

    def pooled_volume(self, parts, addend=None):
        """The (B,C,Y,X,Z) view of the volume, written once; addend (B,C,Y,X) is added broadcast over z in the store."""
        idx, depth, feat, tile_ws = parts
        B, C = depth.shape[0], feat.shape[-1]
        Z, Y, X = self.grid_zyx
        out = torch.empty((B, C, Z, Y, X), dtype=self.out_dtype, device=depth.device)
        if addend is not None and POOL_READ_AHEAD and B * Y * X >= POOL_READ_AHEAD_MIN_QUERIES:
            # round 6: this call comes ~1 ms and ~0.7 GB of intermediate traffic after `parts` was built -- the kernel's gather sources
            # have left the memory-side cache and its dependent gather chains would run at HBM latency (tools/dbg_pool_in_step.py:
            # 260 -> 176 us at BASELINE configs[2], B = 4): read the ~30 MB once, right before.  S3 1.169 -> 1.077 ms there; at B = 1 and at
            # the shipped grid (<= 40 000 queries: the step's intermediates fit the 256 MB cache, and the eager step is host-bound) the extra
            # launch costs 4-40 us: only for large steps (profiles/r06_exp_pool_read_ahead.md)
            _capi.touch(idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts, idx.interval_lengths, depth, feat, tile_ws)
        _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                    idx.interval_lengths, B, C, Z, Y, X, out, tile_ws, self._wo_tile, self.pool_flags,
                                    addend=None if addend is None else addend.contiguous().float())
        return out.permute(0, 1, 3, 4, 2)

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda):
        return None
