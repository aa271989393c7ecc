"""GPU parity tests (run with `-m gpu` on the MI355X): the HIP path, reached through the C ABI of
libfbbev_hip.so, against (a) the CPU oracle on the same seeded inputs, (b) the committed golden
fixtures produced by the real reference Python, (c) the reference's own CUDA kernel compiled for
gfx950 (oracle/_ref), and (d) size-independent properties at BASELINE.json's full sizes.

Bars: bit-exact for every index tensor and for pooled sums taken in the reference's order
(fmaf chain); <= 1e-4 abs (fp32) wherever the summation order legitimately differs."""
import json
import os

import numpy as np
import pytest
import torch

from fb_bev_amd import synthetic as S

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _oracle():
    from oracle import oracle as O
    return O


def _vt(cfg, dev, **kw):
    from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
    return LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, **kw).to(dev)


def _inputs(name, B, aug, dev):
    O = _oracle()
    cfg = S.CONFIGS[name]
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=0, bda_aug=aug)
    coor = ovt.get_lidar_coor(*cam).contiguous()
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    return cfg, ovt, cam, coor, depth, ctx


# ------------------------------------------------------------------ known answer (bev_pool.py:144-175)
def test_known_answer_like_reference_test(dev):
    from fb_bev_amd.bev_pool import bev_pool_v2
    k = json.load(open(os.path.join(G, 'bev_pool_v2_known_answer.json')))
    depth = torch.tensor(k['depth'], device=dev).view(*k['depth_shape']).requires_grad_()
    feat = torch.ones(*k['feat_ones_shape'], device=dev).requires_grad_()
    rd, rf, rb = (torch.tensor(k[n], device=dev).int() for n in ('ranks_depth', 'ranks_feat', 'ranks_bev'))
    kept = torch.ones(rb.shape[0], device=dev, dtype=torch.bool)
    kept[1:] = rb[1:] != rb[:-1]
    starts = torch.where(kept)[0].int()
    lengths = torch.zeros_like(starts)
    lengths[:-1] = starts[1:] - starts[:-1]
    lengths[-1] = rb.shape[0] - starts[-1]
    bev = bev_pool_v2(depth, feat, rd, rf, rb, tuple(k['bev_feat_shape']), starts, lengths)
    loss = bev.sum()
    loss.backward()
    assert abs(loss.item() - k['loss']) < 1e-6
    assert depth.grad.view(-1).cpu().allclose(torch.tensor(k['grad_depth']))
    assert feat.grad.view(-1).cpu().allclose(torch.tensor(k['grad_feat']))


# ------------------------------------------------------------------ ranking: bit-exact index tensors
@pytest.mark.parametrize('tag', ['TINY_B2_aug', 'SMALL_B2_aug'])
def test_rank_build_vs_reference_python_fixture(dev, tag):
    z = np.load(os.path.join(G, f'index_{tag}.npz'))
    cfg = S.CONFIGS[tag.split('_')[0]]
    vt = _vt(cfg, dev)
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(torch.from_numpy(z['coor']).to(dev))
    for got, key in ((rb, 'ranks_bev'), (rd, 'ranks_depth'), (rf, 'ranks_feat'), (st, 'interval_starts'),
                     (ln, 'interval_lengths')):
        assert got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), z[key]), key


@pytest.mark.parametrize('name,B,aug', [('REF', 1, False), ('BL2', 1, False), ('BL2', 2, True),
                                        ('BL1', 1, False), ('BL5', 1, False), ('REF', 3, True)])
def test_rank_build_full_size_bit_exact(dev, name, B, aug):
    cfg, ovt, cam, coor, _, _ = _inputs(name, B, aug, dev)
    exp = ovt.voxel_pooling_prepare_v2(coor)
    got = _vt(cfg, dev).voxel_pooling_prepare_v2(coor.to(dev))
    for g, e, key in zip(got, exp, ('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths')):
        assert torch.equal(g.cpu(), e), (name, key)
    tag = f'{name}_B{B}' + ('_aug' if aug else '')
    stats = json.load(open(os.path.join(G, 'index_stats.json')))
    if tag in stats:  # numbers produced by the real reference Python
        assert (got[0].numel(), got[3].numel(), int(got[4].max())) == (stats[tag]['P'], stats[tag]['I'],
                                                                       stats[tag]['len_max'])


@pytest.mark.parametrize('name,B', [('SMALL', 2), ('REF', 2), ('BL2', 3), ('BL5', 1)])
def test_fused_geometry_rank_build_equals_two_step(dev, name, B):
    """fbbev_lift_rank_build == fbbev_lidar_coor + fbbev_rank_build, bit for bit, and == the oracle on the same coor."""
    cfg, ovt, cam, _, _, _ = _inputs(name, B, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    coor = vt.get_lidar_coor(*cam_g)
    two = vt.build_index(coor).exact()
    one = vt.build_index_from_cams(*cam_g).exact()
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    exp = ovt.voxel_pooling_prepare_v2(coor.cpu())
    for a, e in zip(one, exp):
        assert torch.equal(a.cpu(), e)


def test_fp32_rank_collisions_above_2_24_are_reproduced(dev):
    """the reference evaluates rank = b*ZYX + z*YX + y*X + x in fp32 (view_transformer.py:586-589), so once
    B*X*Y*Z exceeds 2^24 neighbouring voxels of the late samples share a rank.  That merge is part of the reference's
    output; the device-side key evaluation must reproduce it bit for bit (BL2 grid, B=28: 17.9 M voxels)."""
    cfg, ovt, cam, _, _, _ = _inputs('BL2', 28, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    coor_g = vt.get_lidar_coor(*cam_g)                     # the bit-exact contract starts at coor
    coor = coor_g.cpu()
    exp = ovt.voxel_pooling_prepare_v2(coor)
    got = vt.build_index_from_cams(*cam_g).exact()
    two = vt.voxel_pooling_prepare_v2(coor_g)
    for g, t, e, key in zip(got, two, exp, ('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths')):
        assert torch.equal(g.cpu(), e), key
        assert torch.equal(t.cpu(), e), key
    # the quirk is really exercised: exact integer ranks of the kept points differ from the fp32 ones
    Z, Y, X = vt.grid_zyx
    c = coor.view(-1, 3)[exp[1].long()]
    lo, it = ovt.grid_lower_bound, ovt.grid_interval
    v = ((c - lo) / it).long()
    b = exp[1].long() // (coor[0].numel() // 3)
    exact = ((b * Z + v[:, 2]) * Y + v[:, 1]) * X + v[:, 0]
    assert int((exact != exp[0].long()).sum()) > 0


def test_rank_build_with_depth_threshold(dev):
    """BEVDet-era filter kept &= depth > 0.01 (necks/view_transformer.py:552-557) at BL2 size: data-dependent P, no sync."""
    cfg, ovt, cam, coor, depth, _ = _inputs('BL2', 2, True, dev)
    depth = depth.clone()
    depth[torch.rand(depth.shape, generator=torch.Generator().manual_seed(9)) < 0.5] = 0.0
    vt = _vt(cfg, dev)
    idx = vt.build_index(coor.to(dev), depth=depth.to(dev), depth_threshold=0.01)
    c2 = coor.clone()
    c2.view(-1, 3)[~(depth.reshape(-1) > 0.01)] = 1.0e6
    exp = ovt.voxel_pooling_prepare_v2(c2)
    for g, e, key in zip(idx.exact(), exp, ('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths')):
        assert torch.equal(g.cpu(), e), key


def test_rank_build_edge_cases(dev):
    cfg = S.CONFIGS['TINY']
    O = _oracle()
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    vt = _vt(cfg, dev)
    coor = torch.full((1, 1, 2, 2, 3, 3), 1000.0)
    assert vt.voxel_pooling_prepare_v2(coor.to(dev)) == (None,) * 5          # empty frustum
    coor[0, 0, 1, 1, 2] = torch.tensor([-8.5, -8.0, -1.0])                    # trunc toward zero -> voxel 0
    coor[0, 0, 0, 0, 0] = torch.tensor([float('nan'), 0.0, 0.0])
    coor[0, 0, 0, 0, 1] = torch.tensor([7.999, 7.999, 2.999])
    coor[0, 0, 0, 1, 0] = torch.tensor([float('inf'), 0.0, 0.0])
    got = vt.voxel_pooling_prepare_v2(coor.to(dev))
    exp = ovt.voxel_pooling_prepare_v2(coor)
    for g, e in zip(got, exp):
        assert torch.equal(g.cpu(), e)
    # all points in ONE voxel: a single maximal interval
    coor = torch.zeros((1, 2, 3, 4, 5, 3))
    got = vt.voxel_pooling_prepare_v2(coor.to(dev))
    exp = ovt.voxel_pooling_prepare_v2(coor)
    for g, e in zip(got, exp):
        assert torch.equal(g.cpu(), e)
    assert got[4].tolist() == [120]


# get_lidar_coor: observed on an MI355X (profiles/r06_gpu_tests_observed.txt) and the bars set from it (2x observed)
# observed: 3.8e-6 / 9.5e-6 / 1.14e-5 / 1.14e-5 m; voxel flips 0 / 0 / 0 / 35 of 1 786 093 kept points (1.96e-5)
LIDAR_COOR_MAX_ERR = {'SMALL': 8e-6, 'REF': 2e-5, 'BL2': 2.3e-5, 'BL5': 2.3e-5}   # metres
LIDAR_VOXEL_FLIP_FRAC = 4e-5                                                       # kept points whose voxel differs from the oracle's


def test_lidar_coor_close_to_oracle(dev):
    """fbbev_lidar_coor (closed-form 3x3 inverses, one kernel) against the oracle's restatement of view_transformer.py:458-498
    (torch.inverse + matmul chain on the CPU): the largest coordinate difference in metres, and -- what the ranking actually sees --
    how many frustum points land in a DIFFERENT voxel (or flip in / out of the grid) than with the oracle's coordinates.  The index
    contract is pinned AT `coor`, so the index tensors stay bit-exact either way; this measures the end-to-end effect of
    the different 3x3 inverse.  Both numbers are printed; bars = 2x observed."""
    for name in ('SMALL', 'REF', 'BL2', 'BL5'):
        cfg, ovt, cam, coor, _, _ = _inputs(name, 2, True, dev)
        got = _vt(cfg, dev).get_lidar_coor(*[t.to(dev) for t in cam]).cpu()
        err = (got - coor).abs().max().item()
        lo, it, gs = ovt.grid_lower_bound, ovt.grid_interval, ovt.grid_size

        def vox(c):
            v = ((c - lo) / it).long().view(-1, 3)
            inside = ((v >= 0) & (v < gs.long())).all(1)
            return torch.where(inside[:, None], v, torch.full_like(v, -1))
        a, b = vox(got), vox(coor)
        kept = int((b[:, 0] >= 0).sum())
        flips = int((a != b).any(1).sum())
        print(f'get_lidar_coor [{name}]: max|coor - oracle| = {err:.3e} m (fp32 ulp at 50 m: 3.8e-6); {flips} of {kept} kept points '
              f'({flips / max(kept, 1):.2e}) fall in a different voxel than with the torch.inverse coordinates')
        assert err < LIDAR_COOR_MAX_ERR[name], (name, err)
        assert flips <= LIDAR_VOXEL_FLIP_FRAC * kept + 2, (name, flips, kept)


def test_nchw_to_nhwc_kernel(dev):
    from fb_bev_amd import _capi
    for shape in ((2, 6, 80, 16, 44), (1, 6, 64, 64, 176), (3, 2, 33, 5, 7)):
        x = torch.randn(shape, device=dev)
        assert torch.equal(_capi.nchw_to_nhwc(x), x.permute(0, 1, 3, 4, 2).contiguous())


@pytest.mark.parametrize('name,B,dt', [('SMALL', 2, torch.float32), ('REF', 2, torch.float32), ('BL2', 4, torch.float32), ('BL2', 2, torch.bfloat16),
                                       ('BL5', 1, torch.float16)])
def test_lift_splat_fused_one_entry_equals_the_four_call_sequence(dev, name, B, dt):
    """fbbev_lift_splat_fused (view_transformer.py:521-545 + :613-635 as ONE C entry) == the sequence
    fbbev_lift_rank_build -> fbbev_nchw_to_nhwc -> fbbev_pool_tile_index -> fbbev_bev_pool_v2_dense_fwd bit for bit (fp32 and 16-bit
    storage), == the C oracle on the same coor for the fp32 volume, index tensors in the workspace == the oracle's, and the
    camera-keyed form (hit / changed rig) reproduces the same volume."""
    from fb_bev_amd import _capi
    O = _oracle()
    cfg, ovt, cam, _, depth, ctx = _inputs(name, B, True, dev)
    vt = _vt(cfg, dev, out_dtype=dt)
    tv, flags = vt.tiling(cfg.n_cams)
    cam_g = [t.to(dev) for t in cam]
    d_g, c_g = depth.to(dev), ctx.to(dev)
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    four = vt(cam_g, c_g, d_g).permute(0, 1, 4, 2, 3)                       # the module's own 4-call route, (B,C,Z,Y,X)
    N, D, (H, W) = cfg.n_cams, cfg.D, cfg.feat_hw
    dims = (B, N, D, H, W, C, Z, Y, X)
    ws = torch.full((_capi.lift_splat_fused_ws_bytes(*dims),), 0xA5, dtype=torch.uint8, device=dev)
    xs, ys, ds = vt._axes(dev)
    lo, it, gs = vt._grid3()
    out = torch.full((B, C, Z, Y, X), float('nan'), dtype=dt, device=dev)
    _capi.lift_splat_fused(xs, ys, ds, *cam_g, d_g, c_g, lo, it, gs, (Z, Y, X), out, ws, tile_voxels=tv, flags=flags)
    assert not torch.isnan(out.float()).any()
    assert torch.equal(out, four)
    views = _capi.lift_splat_fused_ws_views(ws, *dims)
    P, I = views['counts'].tolist()
    coor = vt.get_lidar_coor(*cam_g).cpu()                                    # contract pinned at the ranking input
    exp_idx = ovt.voxel_pooling_prepare_v2(coor)
    for k, e, n in zip(('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths'), exp_idx, (P, P, P, I, I)):
        assert n == e.numel() and torch.equal(views[k][:n].cpu(), e), k
    if dt == torch.float32:
        rb, rd, rf, st, ln = exp_idx
        exp = O.bev_pool_v2(depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb, ovt.bev_feat_shape(B, C), st, ln)
        assert torch.equal(out.cpu(), exp)
    key = torch.full((_capi.cam_key_words(B, N),), -1, dtype=torch.int32, device=dev)
    state = torch.tensor([0, 0, -1, -1], dtype=torch.int32, device=dev)
    ws2 = torch.full_like(ws, 0x5A)
    cam_b = [t.to(dev) for t in S.camera_rig(cfg, B, seed=9, bda_aug=True)]
    for want, cams in (([0, 1], cam_g), ([1, 1], cam_g), ([0, 2], cam_b), ([0, 3], cam_g)):
        out.fill_(float('nan'))
        _capi.lift_splat_fused(xs, ys, ds, *cams, d_g, c_g, lo, it, gs, (Z, Y, X), out, ws2, tile_voxels=tv, flags=flags, cam_key=key,
                               cache_state=state)
        assert state[:2].tolist() == want and not torch.isnan(out.float()).any()
        if cams is cam_g:
            assert torch.equal(out, four)
    with pytest.raises(_capi.FbbevError):
        _capi.lift_splat_fused(xs, ys, ds, *cam_g, d_g, c_g, lo, it, gs, (Z, Y, X), out, ws[:ws.numel() - 512], tile_voxels=tv, flags=flags)


# ------------------------------------------------------------------ pooling forward
CASES = [('TINY', 2, True), ('SMALL', 2, True), ('REF', 1, False), ('BL2', 2, True), ('BL1', 1, False)]


@pytest.mark.parametrize('name,B,aug', CASES)
def test_pool_forward_bit_exact_vs_oracle_and_reference_kernel(dev, name, B, aug):
    import ref_kernel
    from fb_bev_amd import bev_pool_v2_ext
    O = _oracle()
    cfg, ovt, cam, coor, depth, ctx = _inputs(name, B, aug, dev)
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    shape = ovt.bev_feat_shape(B, cfg.channels)
    d_g, f_g = depth.to(dev), feat.to(dev)
    idx = [t.to(dev) for t in (rd, rf, rb, ln, st)]
    out = torch.zeros(shape, device=dev)
    bev_pool_v2_ext.bev_pool_v2_forward(d_g, f_g, out, *idx)
    exp = O.bev_pool_v2_fwd(depth, feat, rd, rf, rb, shape, st, ln, use_fma=True)
    assert torch.equal(out.cpu(), exp)                      # same fmaf chain as the oracle
    assert (out.cpu() - O.bev_pool_v2_fwd(depth, feat, rd, rf, rb, shape, st, ln, use_fma=False)).abs().max() < 1e-4
    if ref_kernel.available():                              # the reference's own kernel on this GPU
        out_ref = torch.zeros(shape, device=dev)
        ref_kernel.fwd(d_g, f_g, idx[0], idx[1], idx[2], idx[4], idx[3], out_ref)
        assert torch.equal(out, out_ref)


@pytest.mark.parametrize('name,B,aug', CASES)
@pytest.mark.parametrize('tv,flags', [(64, 0), (64, 5), (128, 5), (256, 0x25), (64, 0x425), (128, 0x125), (512, 0x455), (1024, 0x4a1), (256, 0x4f1), (128, 0x24424), (64, 0x22416)])
def test_dense_forward_equals_rows_path(dev, name, B, aug, tv, flags):
    """Fused (B,C,Z,Y,X) kernel == zero-init + rows kernel + permute, bit for bit, and writes every element."""
    from fb_bev_amd import _capi
    from fb_bev_amd.bev_pool import bev_pool_v2
    cfg, ovt, cam, coor, depth, ctx = _inputs(name, B, aug, dev)
    vt = _vt(cfg, dev, tile_voxels=tv)
    idx = vt.build_index(coor.to(dev))
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous().to(dev)
    Z, Y, X = vt.grid_zyx
    out = torch.full((B, cfg.channels, Z, Y, X), float('nan'), device=dev)
    cs = 20 if ((flags >> 4) & 0xF) == 0xF else max(1, (flags >> 4) & 0xF)
    cc = cfg.channels // cs if cfg.channels % (4 * cs) == 0 else cfg.channels
    if (cc * (tv + 4) + 3 * tv + 1024) * 4 > 160 * 1024:
        pytest.skip('tile does not fit the 160 KiB LDS for this channel count (FBBEV_E_UNSUPPORTED by contract)')
    ws = vt._tile_ws(dev, B, tv)
    _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
    _capi.bev_pool_v2_dense_fwd(depth.to(dev), feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,
                                idx.interval_starts, idx.interval_lengths, B, cfg.channels, Z, Y, X, out, ws, tv,
                                flags)
    assert not torch.isnan(out).any()
    rb, rd, rf, st, ln = idx.exact()
    exp = bev_pool_v2(depth.to(dev), feat, rd, rf, rb, (B, Z, Y, X, cfg.channels), st, ln)
    assert torch.equal(out, exp)


def test_fused_module_forward_matches_oracle(dev):
    O = _oracle()
    for name, B in (('SMALL', 2), ('REF', 2)):
        cfg, ovt, cam, _, depth, ctx = _inputs(name, B, True, dev)
        vt = _vt(cfg, dev)
        cam_g = [t.to(dev) for t in cam]
        bev = vt(cam_g, ctx.to(dev), depth.to(dev))
        Z, Y, X = vt.grid_zyx
        assert bev.shape == (B, cfg.channels, Y, X, Z)
        coor = vt.get_lidar_coor(*cam_g).cpu()              # contract pinned at the ranking input
        rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
        exp = O.bev_pool_v2(depth, ctx.permute(0, 1, 3, 4, 2), rd, rf, rb, ovt.bev_feat_shape(B, cfg.channels),
                            st, ln).permute(0, 1, 3, 4, 2)
        assert torch.equal(bev.cpu(), exp)
        ref_shaped = _vt(cfg, dev, fused=False)(cam_g, ctx.to(dev), depth.to(dev))
        assert torch.equal(ref_shaped, bev)


def test_full_size_properties_bl2_batch(dev):
    """BASELINE configs[1] at bench batch: linearity in feat and depth, checksum against index_add."""
    B = 4
    cfg, ovt, cam, coor, depth, ctx = _inputs('BL2', B, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    idx = vt.build_index(vt.get_lidar_coor(*cam_g))
    y1 = vt.lift_splat(idx, d, c)
    y2 = vt.lift_splat(idx, d, 2.0 * c)
    assert torch.equal(y2, 2.0 * y1)                        # exact: power-of-two scaling commutes with fmaf
    y3 = vt.lift_splat(idx, d, c + c.flip(2))
    assert (y3 - (y1 + vt.lift_splat(idx, d, c.flip(2)))).abs().max().item() < 1e-4
    rb, rd, rf, st, ln = idx.exact()
    Z, Y, X = vt.grid_zyx
    feat = c.permute(0, 1, 3, 4, 2).reshape(-1, cfg.channels)
    chk = torch.zeros(B * Z * Y * X, cfg.channels, device=dev)
    chk.index_add_(0, rb.long(), d.reshape(-1)[rd.long(), None] * feat[rf.long()])
    chk = chk.view(B, Z, Y, X, cfg.channels).permute(0, 4, 2, 3, 1)
    assert (y1 - chk).abs().max().item() < 1e-4
    assert int((y1 != 0).any(dim=1).sum()) <= st.numel()    # at most I non-empty voxels


# ------------------------------------------------------------------ pooling backward
@pytest.mark.parametrize('name,B,aug', [('TINY', 2, True), ('SMALL', 2, True), ('REF', 1, False), ('BL2', 2, True)])
def test_pool_backward_vs_oracle_and_reference_kernel(dev, name, B, aug):
    import ref_kernel
    from fb_bev_amd import bev_pool_v2_ext
    O = _oracle()
    cfg, ovt, cam, coor, depth, ctx = _inputs(name, B, aug, dev)
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    shape = ovt.bev_feat_shape(B, cfg.channels)
    og = torch.randn(shape, generator=torch.Generator().manual_seed(7))
    edg, efg = O.bev_pool_v2_bwd(og, depth, feat, rd, rf, rb)
    order = torch.argsort(rf, stable=True)
    rf2, rd2, rb2 = rf[order].contiguous(), rd[order].contiguous(), rb[order].contiguous()
    st2, ln2 = O.intervals_from_sorted(rf2)
    g = lambda t: t.contiguous().to(dev)  # noqa: E731
    dg, fg = torch.zeros_like(depth, device=dev), torch.zeros_like(feat, device=dev)
    bev_pool_v2_ext.bev_pool_v2_backward(g(og), dg, fg, g(depth), g(feat), g(rd2), g(rf2), g(rb2), g(ln2), g(st2))
    assert torch.equal(fg.cpu(), efg)                        # in-order fmaf chain over the interval
    assert (dg.cpu() - edg).abs().max().item() < 1e-4        # wave-tree vs serial sum over channels
    if ref_kernel.available():
        dg_r, fg_r = torch.zeros_like(dg), torch.zeros_like(fg)
        ref_kernel.bwd(g(og), g(depth), g(feat), g(rd2), g(rf2), g(rb2), g(st2), g(ln2), dg_r, fg_r)
        assert torch.equal(fg, fg_r)
        assert (dg - dg_r).abs().max().item() < 1e-4


@pytest.mark.parametrize('name,B,aug,padded', [('TINY', 2, True, False), ('SMALL', 2, True, True), ('REF', 1, False, False),
                                               ('BL2', 2, True, False), ('BL5', 1, True, False)])
def test_fused_dense_backward_sync_free(dev, name, B, aug, padded):
    """fbbev_bev_pool_v2_dense_bwd on the device-side index set of the forward: gradient read in its (B,C,Z,Y,X)
    layout, no re-sort.  feat_grad is bit-exact against the oracle run with each pixel's points in ascending
    depth bin (the order the kernel defines); depth_grad within 1e-4 (lane-tree vs serial channel sum)."""
    from fb_bev_amd import _capi
    O = _oracle()
    cfg, ovt, cam, coor, depth, ctx = _inputs(name, B, aug, dev)
    vt = _vt(cfg, dev)
    idx = vt.build_index_from_cams(*[t.to(dev) for t in cam])
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    g = torch.Generator().manual_seed(7)
    if padded:
        og_full = torch.randn((B, C + 8, Z, Y, X), generator=g).to(dev)
        og = og_full[:, 4:4 + C]
    else:
        og = torch.randn((B, C, Z, Y, X), generator=g).to(dev)
    d_g, f_g = depth.to(dev), feat.to(dev)
    dg = torch.full_like(d_g, float('nan'))
    fg = torch.full_like(f_g, float('nan'))
    N, D, H, W = depth.shape[1:]
    ws = torch.empty(_capi.pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X), dtype=torch.uint8, device=dev)
    _capi.bev_pool_v2_dense_bwd(og, d_g, f_g, idx.ranks_depth, idx.interval_rank, idx.interval_starts, idx.counts,
                                idx.n, (Z, Y, X), dg, fg, ws)
    assert not torch.isnan(dg).any() and not torch.isnan(fg).any()   # written completely, zeros included
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(vt.get_lidar_coor(*[t.to(dev) for t in cam]).cpu())
    o = torch.argsort(rd.long())
    edg, efg = O.bev_pool_v2_bwd(og.cpu().permute(0, 2, 3, 4, 1).contiguous(), depth, feat, rd[o].contiguous(),
                                 rf[o].contiguous(), rb[o].contiguous())
    assert torch.equal(fg.cpu(), efg)
    assert (dg.cpu() - edg).abs().max().item() < 1e-4
    # against the wave-per-interval kernel of the plain op (its order: stable sort of the voxel-sorted arrays)
    e2dg, e2fg = O.bev_pool_v2_bwd(og.cpu().permute(0, 2, 3, 4, 1).contiguous(), depth, feat, rd, rf, rb)
    assert (fg.cpu() - e2fg).abs().max().item() < 1e-4


def test_autograd_backward_has_no_host_sync(dev):
    """The whole training step of the path (index build -> pooling -> backward) enqueues without a host sync."""
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 2, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    d = depth.to(dev).requires_grad_()
    c = ctx.to(dev).requires_grad_()
    w = torch.randn((2, cfg.channels) + tuple(vt.grid_zyx[i] for i in (1, 2, 0)),
                    generator=torch.Generator().manual_seed(11)).to(dev)
    vt(cam_g, c, d).mul(w).sum().backward()          # warm-up: allocations, module caches
    d.grad = c.grad = None
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        vt(cam_g, c, d).mul(w).sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()
    assert d.grad is not None and c.grad is not None and torch.isfinite(d.grad).all()


def test_autograd_through_fused_module(dev):
    O = _oracle()
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 2, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    d = depth.to(dev).requires_grad_()
    c = ctx.to(dev).requires_grad_()
    bev = vt(cam_g, c, d)
    w = torch.randn(bev.shape, generator=torch.Generator().manual_seed(11)).to(dev)
    (bev * w).sum().backward()
    coor = vt.get_lidar_coor(*cam_g).cpu()
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
    og = w.cpu().permute(0, 4, 2, 3, 1).contiguous()         # (B,C,Y,X,Z) -> (B,Z,Y,X,C)
    edg, efg = O.bev_pool_v2_bwd(og, depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb)
    assert (d.grad.cpu() - edg).abs().max().item() < 1e-4
    assert (c.grad.cpu() - efg.permute(0, 1, 4, 2, 3)).abs().max().item() < 1e-4


# ------------------------------------------------------------------ MSDeformAttn
def _msda_cases():
    from test_oracle_msda import CASES as C
    return C + [dict(B=12, Q=2500, M=8, Dh=10, shapes=[[16, 44]], P=8),          # FB-OCC cross-attn value call
                dict(B=12, Q=10000, M=1, Dh=80, shapes=[[16, 44]], P=1),         # depth sampling call
                dict(B=2, Q=10000, M=8, Dh=10, shapes=[[100, 100]], P=4),        # BEV self-attention
                dict(B=6, Q=1000, M=8, Dh=10, shapes=[[32, 88], [16, 44], [8, 22], [4, 11]], P=8)]  # BL3: 4 levels


@pytest.mark.parametrize('ci', range(8))
def test_msda_forward_backward(dev, ci):
    from test_oracle_msda import make_case
    from fb_bev_amd.ms_deform_attn import MultiScaleDeformableAttnFunction_fp32 as F32
    O = _oracle()
    case = _msda_cases()[ci]
    value, ss, ls, loc, w = make_case(**case)
    vg, lg, wg = (t.to(dev).requires_grad_() for t in (value, loc, w))
    out = F32.apply(vg, ss.to(dev), ls.to(dev), lg, wg, 64)
    exp = O.msda_fwd(value, ss, ls, loc, w)
    assert torch.allclose(out.detach().cpu(), exp, atol=1e-4, rtol=1e-5)
    small = case['Q'] <= 100
    if small:
        assert torch.allclose(out.detach().cpu(), O.msda_grid_sample(value, ss, loc, w), atol=1e-4, rtol=1e-5)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(go.to(dev))
    if small or case['Q'] * case['B'] <= 30000:
        egv, egl, egw = O.msda_bwd(value, ss, ls, loc, w, go)
        assert torch.allclose(vg.grad.cpu(), egv, atol=2e-3, rtol=1e-3)   # atomics: order-dependent fp32 sums
        assert torch.allclose(wg.grad.cpu(), egw, atol=1e-4, rtol=1e-4)
        assert torch.allclose(lg.grad.cpu(), egl, atol=2e-3, rtol=1e-3)
    else:  # property at full size: d/dweight of sum(out*go) == sampled value . go  => linear in go
        g1 = wg.grad.clone()
        wg.grad = None; vg.grad = None; lg.grad = None
        out2 = F32.apply(vg, ss.to(dev), ls.to(dev), lg, wg, 64)
        out2.backward(2.0 * go.to(dev))
        assert torch.allclose(wg.grad, 2.0 * g1, atol=1e-5, rtol=1e-5)


def test_msda_kernels_vs_reference_tree_bilinear_vectors(dev):
    """The HIP kernels against tests/golden/msda_bilinear_ref.npz -- outputs of the reference tree's twin of mmcv's bilinear
    device functions (ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-147; tests/test_oracle_msda_ref.py pins the oracle and the
    emulated kernels on them bit for bit).  On the GPU the compiler may contract a*b+c, hence a few-ulp tolerance."""
    import test_oracle_msda_ref as T
    from fb_bev_amd.ms_deform_attn import MultiScaleDeformableAttnFunction_fp32 as F32
    for i in range(0, T.N, 2):
        data, ss, ls, loc, attn, gout, m, c, ch, mask, gi = T._case(i)
        vg, lg, wg = (t.to(dev).requires_grad_() for t in (data, loc, attn))
        out = F32.apply(vg, ss.to(dev), ls.to(dev), lg, wg, 64)
        out.backward(gout.to(dev))
        tol = lambda x: 4e-6 * abs(float(x)) + 1e-7  # noqa: E731
        ref_s = float(T.G['sample'][i]) * float(mask)
        assert abs(out[0, 0, m * ch + c].item() - ref_s) <= tol(ref_s), i
        assert torch.allclose(vg.grad.cpu(), gi, rtol=4e-6, atol=1e-7), i
        assert abs(lg.grad[0, 0, m, 0, 0, 0].item() - float(T.G['grad_w'][i])) <= 4 * tol(T.G['grad_w'][i]) + 1e-6, i
        assert abs(lg.grad[0, 0, m, 0, 0, 1].item() - float(T.G['grad_h'][i])) <= 4 * tol(T.G['grad_h'][i]) + 1e-6, i
        assert abs(wg.grad[0, 0, m, 0, 0].item() - float(T.G['grad_mask'][i])) <= tol(T.G['grad_mask'][i]), i


def test_msda_any_batch_and_im2col_step_ignored(dev):
    """batch 72 with im2col_step 64 fails in mmcv; here any batch works."""
    from test_oracle_msda import make_case
    from fb_bev_amd.ms_deform_attn import ms_deform_attn_forward
    O = _oracle()
    value, ss, ls, loc, w = make_case(B=72, Q=5, M=2, Dh=4, shapes=[[3, 4]], P=2)
    out = ms_deform_attn_forward(value.to(dev), ss.to(dev), ls.to(dev), loc.to(dev), w.to(dev), im2col_step=64)
    assert torch.allclose(out.cpu(), O.msda_fwd(value, ss, ls, loc, w), atol=1e-5)


# ------------------------------------------------------------------ streams / re-entrancy
def test_runs_on_current_stream_and_is_reentrant(dev):
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 2, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    ref = vt(cam_g, c, d).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        outs = [vt(cam_g, c, d) for _ in range(3)]
    s.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


def test_fused_forward_is_hip_graph_capturable(dev):
    """The fused forward path has no host sync and no per-call host state, so the whole view transformation
    (geometry -> ranking -> tile index -> pooling) captures into one hipGraph; replays follow in-place input
    updates (new cameras / features) and match eager execution bit for bit."""
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 2, True, dev)
    vt = _vt(cfg, dev)
    cam_s = [t.to(dev).clone() for t in cam]
    d_s, c_s = depth.to(dev).clone(), ctx.to(dev).clone()
    with torch.no_grad():
        eager0 = vt(cam_s, c_s, d_s).clone()                    # warm-up: allocates the cached workspaces
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            vt(cam_s, c_s, d_s)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_s = vt(cam_s, c_s, d_s)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_s, eager0)
        # new sample in the same static buffers: different rig augmentation + features
        cam2 = S.camera_rig(cfg, 2, seed=7, bda_aug=True)
        depth2, ctx2 = S.depth_and_context(cfg, 2, seed=7)
        for dst, src in zip(cam_s, cam2):
            dst.copy_(src)
        d_s.copy_(depth2); c_s.copy_(ctx2)
        g.replay()
        torch.cuda.synchronize()
        eager1 = vt([t.to(dev) for t in cam2], ctx2.to(dev), depth2.to(dev))
        assert torch.equal(out_s, eager1)
        assert not torch.equal(eager0, eager1)


def test_accelerate_is_a_camera_keyed_cache(dev):
    """accelerate=True (view_transformer.py:607-643, disabled by `assert False` at :628 in the reference because nothing
    invalidates it): the index set is keyed on the camera tensors ON THE DEVICE.  Same rig -> the rank build is skipped
    (no rebuild, no host sync); a changed bda / post_rots -> rebuilt, bit-exact with a per-call build."""
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 2, True, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    plain = _vt(cfg, dev)
    ref = plain(cam_g, c, d)
    vt = _vt(cfg, dev, accelerate=True)
    with torch.no_grad():
        a = vt(cam_g, c, d)
        assert vt.index_builds() == 1
        torch.cuda.set_sync_debug_mode('error')                 # the cached call path has no host sync
        try:
            b = vt([t.clone() for t in cam_g], 2.0 * c, d)      # equal VALUES in other buffers: still a hit
        finally:
            torch.cuda.set_sync_debug_mode('default')
        assert vt.index_builds() == 1                           # no rebuild on the second call
        assert torch.equal(a, ref) and torch.equal(b, 2.0 * ref)
        cam2 = [t.clone() for t in cam_g]
        rz = torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]], device=dev)
        cam2[5][1] = cam2[5][1] @ rz                            # another BEV augmentation for sample 1
        e = vt(cam2, c, d)
        assert vt.index_builds() == 2
        assert torch.equal(e, plain(cam2, c, d)) and not torch.equal(e, a)
        cam3 = [t.clone() for t in cam2]
        cam3[3][0, 2] = cam3[3][0, 2] * 1.1                     # image-space augmentation of one camera (post_rots)
        f = vt(cam3, c, d)
        assert vt.index_builds() == 3 and torch.equal(f, plain(cam3, c, d))
        g = vt(cam3, c, d)
        assert vt.index_builds() == 3 and torch.equal(g, f)
        # write-once route (FBViewTransform) shares the cache
        parts = vt.pooling_inputs(cam3, c, d)
        assert vt.index_builds() == 3
        assert torch.equal(vt.pooled_volume(parts), f)
    # under autograd the persistent buffers are not used (an earlier graph must keep its own index set)
    dg = d.clone().requires_grad_()
    h = vt(cam_g, c, dg)
    assert vt.index_builds() == 3 and torch.equal(h.detach(), ref)
    h.sum().backward()
    assert dg.grad is not None
    # reference-API attributes
    vt.pre_compute(cam_g)
    assert vt.ranks_bev.dtype == torch.int32 and vt.interval_starts.numel() > 0 and not vt.initial_flag


@pytest.mark.parametrize('out_dtype', [torch.float32, torch.bfloat16])
def test_cached_tile_tables_survive_mixed_routes(dev, out_dtype):
    """the tile tables of accelerate=True belong to the cached index set, one per tile size, each
    gated on the build it was made for.  A fixed validation rig alternating with grad-enabled calls of ANOTHER rig (same
    B), and the two routes of a 16-bit volume (lift_splat: doubled 16-bit tile; write-once: fp32 tile) interleaved
    across rebuilds, must always pool with the table of the current index set."""
    # 16-bit volumes need (Y*X) % 8 == 0 for the fused kernels (SMALL is 50x50 and would take the reference-shaped path)
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL' if out_dtype == torch.float32 else 'REF', 2, True, dev)
    rz = torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]], device=dev)
    cam_a = [t.to(dev) for t in cam]
    cam_b = [t.clone() for t in cam_a]
    cam_b[5][0] = cam_b[5][0] @ rz
    cam_b[5][1] = cam_b[5][1] @ rz @ rz
    d, c = depth.to(dev), ctx.to(dev)
    plain = _vt(cfg, dev, out_dtype=out_dtype)
    ref_a, ref_b = plain(cam_a, c, d), plain(cam_b, c, d)
    assert not torch.equal(ref_a, ref_b)
    vt = _vt(cfg, dev, accelerate=True, out_dtype=out_dtype)
    assert vt._fused_supported(cfg.channels)
    with torch.no_grad():
        assert torch.equal(vt(cam_a, c, d), ref_a)                       # build 1, table of lift_splat's tile
    # training-style call of another rig, same B: its per-call index set and table must not leak into the cache
    dg = d.clone().requires_grad_()
    assert torch.equal(vt(cam_b, c, dg).detach(), ref_b)
    with torch.no_grad():
        assert torch.equal(vt(cam_a, c, d), ref_a)                       # hit: cached table still the one of rig a
        assert vt.index_builds() == 1
        # the write-once route (another tile size for 16-bit volumes) first used on a HIT: its table is new -> built
        assert torch.equal(vt.pooled_volume(vt.pooling_inputs(cam_a, c, d)), ref_a)
        assert vt.index_builds() == 1
        # rebuild through the write-once route only, then a hit through lift_splat: ITS table is one build behind
        assert torch.equal(vt.pooled_volume(vt.pooling_inputs(cam_b, c, d)), ref_b)
        assert vt.index_builds() == 2
        assert torch.equal(vt(cam_b, c, d), ref_b)
        assert vt.index_builds() == 2
        # and the other way round
        assert torch.equal(vt(cam_a, c, d), ref_a)
        assert vt.index_builds() == 3
        assert torch.equal(vt.pooled_volume(vt.pooling_inputs(cam_a, c, d)), ref_a)
        zm = vt.pooled_zmean(vt.pooling_inputs(cam_a, c, d))
        assert vt.index_builds() == 3
    full = _vt(cfg, dev)(cam_a, c, d)                                    # fp32 volume (B,C,Y,X,Z)
    assert (zm - full.mean(-1)).abs().max().item() <= 1e-5


def test_cached_index_build_is_graph_capturable(dev):
    """The camera-key compare + early-out chain contains no host decision: captured once, replayed with the same rig
    (skip) and with a new rig written into the captured input buffers (rebuild)."""
    cfg, ovt, cam, _, depth, ctx = _inputs('SMALL', 1, True, dev)
    vt = _vt(cfg, dev, accelerate=True)
    plain = _vt(cfg, dev)
    cam_g = [t.to(dev).clone() for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    with torch.no_grad():
        vt(cam_g, c, d)                                         # warm-up: allocations + first build
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            vt(cam_g, c, d)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = vt(cam_g, c, d)
        g.replay()
        assert torch.equal(out, plain(cam_g, c, d))
        builds = vt.index_builds()
        new = [t.clone() for t in cam_g]
        new[5][0] = new[5][0] @ torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]], device=dev)
        for dst, src in zip(cam_g, new):
            dst.copy_(src)
        g.replay()
        assert vt.index_builds() == builds + 1
        assert torch.equal(out, plain(cam_g, c, d))


def test_dense_forward_stress_grid_bl5(dev):
    """BASELINE configs[4] grid (400x400x16 = 2.56 M voxels/sample, D=118): fused dense == rows path."""
    from fb_bev_amd.bev_pool import bev_pool_v2
    cfg, ovt, cam, _, depth, ctx = _inputs('BL5', 1, True, dev)
    vt = _vt(cfg, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    out = vt(cam_g, c, d)                                               # (B,C,Y,X,Z) view of (B,C,Z,Y,X)
    Z, Y, X = vt.grid_zyx
    assert out.shape == (1, cfg.channels, Y, X, Z)
    rb, rd, rf, st, ln = vt.build_index_from_cams(*cam_g).exact()
    exp = bev_pool_v2(d, c.permute(0, 1, 3, 4, 2), rd, rf, rb, (1, Z, Y, X, cfg.channels), st, ln)
    assert torch.equal(out.permute(0, 1, 4, 2, 3), exp)
    stats = json.load(open(os.path.join(G, 'index_stats.json')))['BL5_B1']
    assert abs(rb.numel() - stats['P']) < 0.2 * stats['P']             # augmented rig: same order of magnitude


@pytest.mark.parametrize('name,B', [('BL2', 16), ('BL5', 1)])
def test_full_size_pooled_volume_vs_oracle_and_reference_kernel(dev, name, B):
    """VERDICT r2 (untested sizes): the pooled VOLUME at the sizes that are timed -- BASELINE configs[1] at the bench batch
    (B = 16, the tile size and pool flags `bench.py` uses: vt.tiling()) and the BASELINE configs[4] grid (400x400x16, D = 118)
    -- through the bench's own call sequence (fbbev_lift_rank_build -> nchw_to_nhwc -> fbbev_pool_tile_index ->
    fbbev_bev_pool_v2_dense_fwd), bit for bit against (1) the loop-exact C oracle of bev_pool_cuda.cu:18-45 (OpenMP over
    intervals) and (2) the reference's own kernel compiled for gfx950 (oracle/_ref) on this GPU.  The index tensors are
    compared with the oracle's ranking first."""
    import ref_kernel
    from fb_bev_amd import _capi
    O = _oracle()
    cfg = S.CONFIGS[name]
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=0, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    vt = _vt(cfg, dev)
    tv, flags = vt.tiling(cfg.n_cams)
    if name == 'BL2':
        assert (tv, flags) == (128, 0x24424)                    # what BENCH_r02.json's config block reports
    cam_g = [t.to(dev) for t in cam]
    d_g, c_g = depth.to(dev), ctx.to(dev)
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    idx = vt.build_index_from_cams(*cam_g)
    feat_g = _capi.nchw_to_nhwc(c_g)
    tws = vt._tile_ws(dev, B, tv)
    _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, tws, tv)
    out = torch.full((B, C, Z, Y, X), float('nan'), device=dev)
    _capi.bev_pool_v2_dense_fwd(d_g, feat_g, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                idx.interval_lengths, B, C, Z, Y, X, out, tws, tv, flags)
    rb, rd, rf, st, ln = idx.exact()
    # index tensors == the oracle's ranking of the same coor bits (contract pinned at coor)
    coor = vt.get_lidar_coor(*cam_g).cpu()
    erb, erd, erf, est, eln = ovt.voxel_pooling_prepare_v2(coor)
    for got, exp in ((rb, erb), (rd, erd), (rf, erf), (st, est), (ln, eln)):
        assert torch.equal(got.cpu(), exp)
    # (2) the reference's own kernel on this GPU: (B,Z,Y,X,C), pre-zeroed, then the permute of bev_pool.py:88 as a view
    if ref_kernel.available():
        out_ref = torch.zeros((B, Z, Y, X, C), device=dev)
        ref_kernel.fwd(d_g, feat_g, rd.contiguous(), rf.contiguous(), rb.contiguous(), st.contiguous(), ln.contiguous(), out_ref)
        assert torch.equal(out, out_ref.permute(0, 4, 1, 2, 3))
        del out_ref
    # (1) the C oracle (same fmaf chain), sample by sample to bound host memory
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    exp = O.bev_pool_v2_fwd(depth, feat, erd, erf, erb, ovt.bev_feat_shape(B, C), est, eln, use_fma=True)     # (B,Z,Y,X,C)
    for b in range(B):
        assert torch.equal(out[b].cpu(), exp[b].permute(3, 0, 1, 2)), b


@pytest.mark.parametrize('name,B,dt', [('BL2', 2, torch.bfloat16), ('BL2', 1, torch.float16), ('REF', 2, torch.bfloat16)])
def test_16bit_storage_equals_rounded_fp32_volume(dev, name, B, dt):
    """FBBEV_POOL_OUT_BF16 / OUT_F16: the same fp32 in-order sums, rounded once (nearest-even) at the store."""
    cfg, ovt, cam, _, depth, ctx = _inputs(name, B, True, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    full = _vt(cfg, dev)(cam_g, c, d)
    half = _vt(cfg, dev, out_dtype=dt)(cam_g, c, d)
    assert half.dtype == dt and half.shape == full.shape
    assert torch.equal(half.contiguous().view(torch.int16), full.to(dt).contiguous().view(torch.int16))
    # autograd still works through the 16-bit volume (gradient converted to fp32 for the backward kernels)
    d2, c2 = d.clone().requires_grad_(), c.clone().requires_grad_()
    _vt(cfg, dev, out_dtype=dt)(cam_g, c2, d2).float().sum().backward()
    d3, c3 = d.clone().requires_grad_(), c.clone().requires_grad_()
    _vt(cfg, dev)(cam_g, c3, d3).sum().backward()
    assert torch.allclose(d2.grad, d3.grad, atol=1e-4) and torch.allclose(c2.grad, c3.grad, atol=1e-4)


@pytest.mark.parametrize('name,B', [('BL1', 1), ('REF', 2)])
def test_pool_tolerance_mode_on_gpu(dev, name, B):
    """VERDICT r3 (weak 1-ii): `pool_tolerance=True` (FBBEV_POOL_SPLIT_LONG) had no GPU test.  BASELINE configs[0] (BL1: 64x176
    features, intervals of up to ~3 900 points) and the shipped grid (REF), through the module: the pooled volume is within
    1e-4 of the volume's scale of the loop-exact C oracle (bev_pool_cuda.cu:18-45, the serial fmaf chain), bit-identical run to
    run, bit-identical to the default (serial) kernel in every voxel whose interval has <= 32 points, and the long intervals
    really took the other summation order."""
    O = _oracle()
    cfg, ovt, cam, _, depth, ctx = _inputs(name, B, True, dev)
    cam_g = [t.to(dev) for t in cam]
    d, c = depth.to(dev), ctx.to(dev)
    exact_vt = _vt(cfg, dev)
    tol_vt = _vt(cfg, dev, pool_tolerance=True)
    tv, fl = tol_vt.tiling(cfg.n_cams)
    from fb_bev_amd import _capi
    assert fl & _capi.POOL_SPLIT_LONG, 'the tolerance flag did not reach the tiling of this shape'
    base = exact_vt(cam_g, c, d)
    tol = tol_vt(cam_g, c, d)
    again = tol_vt(cam_g, c, d)
    assert torch.equal(tol, again)                                      # deterministic
    Z, Y, X = exact_vt.grid_zyx
    C = cfg.channels
    coor = exact_vt.get_lidar_coor(*cam_g).cpu()
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    exp = O.bev_pool_v2_fwd(depth, feat, rd, rf, rb, ovt.bev_feat_shape(B, C), st, ln, use_fma=True)          # (B,Z,Y,X,C)
    got = tol.permute(0, 4, 2, 3, 1).cpu()                              # (B,C,Y,X,Z) view -> (B,Z,Y,X,C)
    assert torch.equal(base.permute(0, 4, 2, 3, 1).cpu(), exp)          # the default kernel IS the oracle, bit for bit
    scale = exp.abs().max().item()
    err = (got - exp).abs().max().item()
    print(f'pool tolerance mode {name} B={B}: max|err| = {err:.3e} = {err / scale:.2e} of the volume scale {scale:.3f}; '
          f'longest interval {int(ln.max())}')
    assert err <= 1e-4 * scale
    long_vox = rb[st.long()][ln > 32].long()
    assert long_vox.numel() > 0
    flat_g, flat_e = got.reshape(-1, C), exp.reshape(-1, C)
    short = torch.ones(flat_e.shape[0], dtype=torch.bool)
    short[long_vox] = False
    assert torch.equal(flat_g[short], flat_e[short])                    # short intervals and empty voxels: the same bits
    assert not torch.equal(flat_g[long_vox], flat_e[long_vox])          # long intervals: another (deterministic) order


@pytest.mark.gpu
def test_volume_zreduce_and_the_training_path_functions_match_torch():
    """fbbev_volume_zreduce (Z-mean / Z-sum of a materialised volume through its (B,C,Y,X,Z) view) and the two autograd functions
    the training path builds on it (fb_view_transform._ZMean / _ReAdd) against the ATen expressions they replace
    (fbocc.py:359, 365-366), values and gradients."""
    from fb_bev_amd import _capi
    from fb_bev_amd.fb_view_transform import _ReAdd, _ZMean
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    vol = torch.randn(2, 5, 8, 12, 20, generator=g).to(dev)                       # (B, C, Z, Y, X) memory
    view = vol.permute(0, 1, 3, 4, 2)                                             # the module's (B, C, Y, X, Z) view
    assert _capi.volume_zreduce_supported(view) and not _capi.volume_zreduce_supported(view.contiguous())
    assert torch.allclose(_capi.volume_zreduce(view, 8), view.mean(-1), rtol=1e-6, atol=1e-6)
    assert torch.allclose(_capi.volume_zreduce(view, 1.0), view.sum(-1), rtol=1e-6, atol=1e-5)
    # a gradient that arrives CONTIGUOUS in the output's (B,C,Y,X,Z) shape: Z-innermost reduction and the re-layout for the pooling backward
    zl = view.contiguous()
    assert _capi.volume_zlast_supported(zl) and not _capi.volume_zlast_supported(view)
    assert torch.allclose(_capi.volume_zreduce_inner(zl, 1.0), zl.sum(-1), rtol=1e-6, atol=1e-5)
    assert torch.equal(_capi.volume_z_to_front(zl), vol)
    w = torch.randn(view.shape, generator=g).to(dev)
    ref = torch.randn(2, 5, 12, 20, generator=g).to(dev)
    outs = []
    for fast in (True, False):
        v = vol.clone().requires_grad_()
        r = ref.clone().requires_grad_()
        vv = v.permute(0, 1, 3, 4, 2)
        zm = _ZMean.apply(vv) if fast else vv.mean(-1)
        out = _ReAdd.apply(r * zm, vv) if fast else (r * zm)[..., None] + vv
        (out * w).sum().backward()
        outs.append((out.detach(), v.grad, r.grad))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
