"""Kernel-logic tests on the CPU fiber emulator (tests/emu): the same kernel source and C-ABI
launch code as the product, executed at tiny sizes and compared with the oracle.  These do NOT
replace the `-m gpu` parity tests (memory model, wave64 hardware behaviour and performance are only
visible on the MI355X); they catch indexing / scan / tiling mistakes before a GPU call is spent."""
import os
import sys

import numpy as np
import pytest
import torch

from fb_bev_amd import synthetic as S
from oracle import oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))
import emu_capi as E  # noqa: E402


def _grid3(vt):
    return vt.grid_lower_bound.tolist(), vt.grid_interval.tolist(), vt.grid_size.tolist()


def _case(name, B, aug=True):
    cfg = name if isinstance(name, S.PathConfig) else S.CONFIGS[name]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=0, bda_aug=aug)
    coor = vt.get_lidar_coor(*cam).contiguous()
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    return cfg, vt, coor, depth, feat


@pytest.mark.parametrize('name,B', [('TINY', 2), ('TINY', 1)])
def test_rank_build_bit_exact(name, B):
    cfg, vt, coor, _, _ = _case(name, B)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    P, I = counts.tolist()
    assert (P, I) == (erb.numel(), est.numel())
    assert torch.equal(rb[:P], erb) and torch.equal(rd[:P], erd) and torch.equal(rf[:P], erf)
    assert torch.equal(st[:I], est) and torch.equal(ln[:I], eln)
    assert torch.equal(ir[:I], erb[est.long()])


def test_rank_build_golden_coor():
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'index_TINY_B2_aug.npz'))
    cfg = S.CONFIGS['TINY']
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(torch.from_numpy(z['coor']), *_grid3(vt))
    P, I = counts.tolist()
    assert np.array_equal(rb[:P].numpy(), z['ranks_bev']) and np.array_equal(rd[:P].numpy(), z['ranks_depth'])
    assert np.array_equal(rf[:P].numpy(), z['ranks_feat'])
    assert np.array_equal(st[:I].numpy(), z['interval_starts']) and np.array_equal(ln[:I].numpy(), z['interval_lengths'])


def test_rank_build_empty_and_edges():
    cfg = S.CONFIGS['TINY']
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    coor = torch.full((1, 1, 2, 2, 3, 3), 1000.0)                # everything outside
    *_, counts = E.rank_build(coor, *_grid3(vt))
    assert counts.tolist() == [0, 0]
    coor[0, 0, 1, 1, 2] = torch.tensor([-8.5, -8.0, -1.0])      # (-1,0) voxel coord truncates to 0
    coor[0, 0, 0, 0, 0] = torch.tensor([float('nan'), 0.0, 0.0])
    coor[0, 0, 0, 0, 1] = torch.tensor([7.999, 7.999, 2.999])   # last voxel
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    assert counts.tolist() == [2, 2]
    assert torch.equal(rb[:2], erb) and torch.equal(rd[:2], erd) and torch.equal(rf[:2], erf)


@pytest.mark.parametrize('name,B', [('TINY', 2)])
def test_pool_fwd_rows_and_bwd(name, B):
    cfg, vt, coor, depth, feat = _case(name, B)
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    shape = vt.bev_feat_shape(B, cfg.channels)
    out = torch.zeros(shape)
    E.pool_fwd(depth, feat, out, rd, rf, rb, st, ln)
    exp = O.bev_pool_v2_fwd(depth, feat, rd, rf, rb, shape, st, ln, use_fma=True)
    assert torch.equal(out, exp)                                  # same fmaf chain -> bit-exact
    # backward: intervals over ranks_feat
    order = torch.argsort(rf, stable=True)
    rf2, rd2, rb2 = rf[order].contiguous(), rd[order].contiguous(), rb[order].contiguous()
    st2, ln2 = O.intervals_from_sorted(rf2)
    og = torch.randn(shape, generator=torch.Generator().manual_seed(3))
    dg, fg = torch.zeros_like(depth), torch.zeros_like(feat)
    E.pool_bwd(og, dg, fg, depth, feat, rd2, rf2, rb2, st2.contiguous(), ln2.contiguous())
    edg, efg = O.bev_pool_v2_bwd(og, depth, feat, rd, rf, rb)
    assert torch.equal(fg, efg)                                   # in-order fmaf chain
    assert torch.allclose(dg, edg, atol=1e-5, rtol=1e-5)          # wave-tree vs serial channel sum


@pytest.mark.parametrize('tv,flags', [(64, 0), (128, 4), (256, 0x24), (64, 0x421), (128, 0x125), (512, 0x405), (1024, 0x24)])
def test_pool_dense_matches_oracle(tv, flags):
    cfg, vt, coor, depth, feat = _case('TINY', 2)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    B, Z, Y, X, C = vt.bev_feat_shape(2, cfg.channels)
    code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    assert code == 0
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    exp = O.bev_pool_v2(depth, feat, erd, erf, erb, (B, Z, Y, X, C), est, eln, use_fma=True)
    assert not torch.isnan(out).any()                             # every element written exactly once
    assert torch.equal(out, exp)


@pytest.mark.parametrize('tv,flags', [(128, 4), (128, 0x24), (256, 0), (128, 0x125)])
def test_pool_dense_tile_with_more_points_than_the_staged_index_window(tv, flags):
    """A tile whose points do not fit the FBBEV_NP_STAGE = 512 index pairs staged in LDS: intervals before, ACROSS (in the
    4-point batches and in the one-point tail) and beyond the window take their indices from LDS / global memory / both.
    Hand-built index set on a 1 x 8 x 32 grid (two 128-voxel tiles per plane), lengths 1..37; bit-exact vs the oracle."""
    g = torch.Generator().manual_seed(tv + flags)
    B, Z, Y, X, C = 1, 2, 8, 32, 16
    YX = Y * X
    lens = torch.randint(1, 12, (B * Z * YX,), generator=g)
    lens[torch.randperm(lens.numel(), generator=g)[:40]] = torch.randint(20, 38, (40,), generator=g)   # long intervals too
    lens[torch.randperm(lens.numel(), generator=g)[:60]] = 0                                           # and empty voxels
    ranks = torch.repeat_interleave(torch.arange(B * Z * YX), lens).int()       # sorted voxel rank of every point
    P = ranks.numel()
    assert P > 2500                                                              # ~ 700 points per 128-voxel tile
    n_src = 900
    rd = torch.randint(0, n_src, (P,), generator=g).int()
    rf = torch.randint(0, n_src // 4, (P,), generator=g).int()
    depth = torch.randn(1, 1, 1, 1, n_src, generator=g)
    feat = torch.randn(1, 1, 1, n_src // 4, C, generator=g)
    st, ln = O.intervals_from_sorted(ranks)
    st, ln = st.int().contiguous(), ln.int().contiguous()
    ir = ranks[st.long()].contiguous()
    counts = torch.tensor([P, st.numel()], dtype=torch.int32)
    code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    assert code == 0
    exp = O.bev_pool_v2(depth, feat, rd, rf, ranks, (B, Z, Y, X, C), st, ln, use_fma=True)
    assert not torch.isnan(out).any()
    assert torch.equal(out, exp)
    # the case is what the docstring says: some interval starts inside the window and ends beyond it, in every tile
    tile = (ranks.long() // YX) * (YX // tv) + (ranks.long() % YX) // tv
    first = torch.full((int(tile.max()) + 1,), P, dtype=torch.long).scatter_reduce_(0, tile, torch.arange(P), 'amin')
    rel = st.long() - first[tile[st.long()]]
    assert ((rel < 512) & (rel + ln.long() > 512)).any() and (rel >= 512).any()


def test_pool_dense_rejects_unsupported():
    cfg, vt, coor, depth, feat = _case('TINY', 1)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    code, _ = E.pool_dense(depth, feat[..., :6].contiguous(), rd, rf, ir, st, ln, counts, st.numel(),
                           1, 6, 4, 16, 16, 128)
    assert code == -2                                             # FBBEV_E_UNSUPPORTED (C % 4 != 0)


def test_msda_fwd_bwd_emulated():
    from test_oracle_msda import CASES, make_case
    for case in CASES[:3]:
        value, ss, ls, loc, w = make_case(**case)
        out = E.msda_fwd(value, ss, ls, loc, w)
        assert torch.allclose(out, O.msda_fwd(value, ss, ls, loc, w), atol=1e-6, rtol=1e-6)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
        gv, gl, gw = E.msda_bwd(value, ss, ls, loc, w, go)
        egv, egl, egw = O.msda_bwd(value, ss, ls, loc, w, go)
        assert torch.allclose(gv, egv, atol=1e-5, rtol=1e-5)
        assert torch.allclose(gl, egl, atol=1e-4, rtol=1e-4)
        assert torch.allclose(gw, egw, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('lds_kb', [0, 8])
def test_msda_bwd_band_binned_fixed_point_emulated(lds_kb, monkeypatch):
    """fbbev_msda_bwd_ws (query bins per band of token rows, 64-bit fixed-point LDS planes, every token written once) against
    the oracle and against the atomic kernel: random sampling all over the levels (query spans = everything), several
    bands per level with the small LDS budget, Dh = 59 has no plan; and a BEV-like case -- raster-ordered queries sampling
    around their own cell -- where the spans must be short and the result identical."""
    from test_oracle_msda import CASES, make_case
    if lds_kb:
        monkeypatch.setenv('FBBEV_MSDA_BWD_LDS_KB', str(lds_kb))
    for case in CASES:
        value, ss, ls, loc, w = make_case(**case)
        go = torch.randn(value.shape[0], loc.shape[1], value.shape[2] * value.shape[3], generator=torch.Generator().manual_seed(1))
        got = E.msda_bwd_ws(value, ss, ls, loc, w, go, case['shapes'])
        if case['Dh'] == 59:
            assert got is None
            continue
        gv, gl, gw = got
        egv, egl, egw = O.msda_bwd(value, ss, ls, loc, w, go)
        assert not torch.isnan(gv).any()
        assert torch.allclose(gv, egv, atol=1e-5, rtol=1e-5)
        assert torch.allclose(gl, egl, atol=1e-4, rtol=1e-4)
        assert torch.allclose(gw, egw, atol=1e-5, rtol=1e-5)
    # BEV self-attention shape: 24 x 20 cells, queries in raster order, 4 points within +-2.5 cells of the query's cell
    g = torch.Generator().manual_seed(9)
    H, W, M, Dh, P, B = 24, 20, 2, 10, 4, 2
    ss = torch.tensor([[H, W]], dtype=torch.int64); ls = torch.zeros(1, dtype=torch.int64)
    value = torch.randn(B, H * W, M, Dh, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    ref = torch.stack([(xs.flatten() + 0.5) / W, (ys.flatten() + 0.5) / H], -1)               # (Q, 2) as (x, y)
    off = (torch.rand(B, H * W, M, 1, P, 2, generator=g) - 0.5) * 5.0 / torch.tensor([W, H])
    loc = (ref[None, :, None, None, None, :] + off).contiguous()
    w = torch.rand(B, H * W, M, 1, P, generator=g).softmax(-1).contiguous()
    go = torch.randn(B, H * W, M * Dh, generator=g) * 1e3                                       # a large gradient scale
    gv, gl, gw = E.msda_bwd_ws(value, ss, ls, loc, w, go, [[H, W]])
    egv, egl, egw = O.msda_bwd(value, ss, ls, loc, w, go)
    assert not torch.isnan(gv).any()
    assert (gv - egv).abs().max() <= 2e-6 * egv.abs().max()
    assert torch.allclose(gl, egl, atol=1e-1, rtol=1e-4) and torch.allclose(gw, egw, atol=1e-2, rtol=1e-5)
    # bit-reproducible: a second run gives the same bits (integer sums), and a poisoned gradient poisons only its band span
    gv2, _, _ = E.msda_bwd_ws(value, ss, ls, loc, w, go, [[H, W]])
    assert torch.equal(gv, gv2)


def test_pool_dense_partial_tiles_small_config():
    """SMALL: 50x50x8 grid (YX=2500 is not a multiple of 64/128), C=20 (5 lanes per interval)."""
    cfg, vt, coor, depth, feat = _case('SMALL', 1)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    B, Z, Y, X, C = vt.bev_feat_shape(1, cfg.channels)
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    P, I = counts.tolist()
    assert (P, I) == (erb.numel(), est.numel())
    assert torch.equal(rb[:P], erb) and torch.equal(rd[:P], erd) and torch.equal(st[:I], est)
    exp = O.bev_pool_v2(depth, feat, erd, erf, erb, (B, Z, Y, X, C), est, eln, use_fma=True)
    for tv, flags in ((64, 0), (128, 0x24), (64, 0x14)):   # C=20: cpl8 falls back to 4; csplit 2 -> 10 ch
        code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
        assert code == 0 and not torch.isnan(out).any()
        assert torch.equal(out, exp)


def test_pool_dense_tolerance_mode_splits_long_intervals_emulated():
    """FBBEV_POOL_SPLIT_LONG (0x2000000): intervals of more than 32 points are summed by all lane groups of the workgroup
    (contiguous chunks, partial sums added in group order).  Hand-made index tensors with interval lengths 1 .. 3 900 (the
    BASELINE configs[0] maximum is 3 894; more points per tile than the 512 staged indices): the result is within 1e-4
    relative of the serial chain (the default kernel == the C oracle, bit for bit), identical run to run, and identical
    to the default wherever no interval is long."""
    g = torch.Generator().manual_seed(5)
    B, Z, Y, X, C = 1, 2, 8, 32, 80                       # YX = 256: two 128-voxel tiles (four 64-voxel tiles) per plane
    N, D, H, W = 2, 6, 8, 16
    lens = {3: 1, 10: 33, 11: 32, 40: 700, 41: 5, 130: 3900, 200: 64, 255: 2, 256 + 7: 129, 256 + 200: 31}
    depth = (torch.randn(B, N, D, H, W, generator=g) * 3).softmax(2).contiguous()
    feat = torch.randn(B, N, H, W, C, generator=g)
    ir = torch.tensor(sorted(lens), dtype=torch.int32)
    ln = torch.tensor([lens[int(v)] for v in ir], dtype=torch.int32)
    st = (torch.cumsum(ln, 0) - ln).int()
    P = int(ln.sum())
    rd = torch.randint(0, depth.numel(), (P,), generator=g, dtype=torch.int32)
    rf = torch.randint(0, B * N * H * W, (P,), generator=g, dtype=torch.int32)
    rb = torch.repeat_interleave(ir, ln.long()).int()
    counts = torch.tensor([P, ir.numel()], dtype=torch.int32)
    exp = O.bev_pool_v2(depth, feat, rd, rf, rb, (B, Z, Y, X, C), st, ln, use_fma=True)
    for tv, flags in ((128, 0x24424), (64, 0x20414), (64, 0x20404)):        # bench default (csplit 2, 8 ch/lane); dense-grid default; 4 ch/lane
        code, base = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, ir.numel(), B, C, Z, Y, X, tv, flags)
        assert code == 0 and torch.equal(base, exp)
        code, tol = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, ir.numel(), B, C, Z, Y, X, tv, flags | 0x2000000)
        assert code == 0 and not torch.isnan(tol).any()
        code, again = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, ir.numel(), B, C, Z, Y, X, tv, flags | 0x2000000)
        assert torch.equal(tol, again)                                      # deterministic
        scale = exp.abs().max().item()
        assert (tol - exp).abs().max().item() <= 1e-4 * scale               # north_star's bar for pooled features
        assert not torch.equal(tol, exp)                                    # the long intervals really took another order
        vox = exp.permute(0, 2, 3, 4, 1).reshape(-1, C)
        tvx = tol.permute(0, 2, 3, 4, 1).reshape(-1, C)
        short = [int(v) for v in ir if lens[int(v)] <= 32]
        assert all(torch.equal(vox[v], tvx[v]) for v in short)              # short intervals: the same serial chain
        empty = torch.ones(vox.shape[0], dtype=torch.bool); empty[ir.long()] = False
        assert not tvx[empty].any()
    # outside the mode's instantiations (16-bit volume): refused, not silently ignored
    code, _ = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, ir.numel(), B, C, Z, Y, X, 128, 0x24424 | 0x2000000 | 0x800000)
    assert code == -4 or code != 0


def test_lidar_coor_emulated():
    for name in ('TINY', 'SMALL'):
        cfg = S.CONFIGS[name]
        vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
        cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
        exp = vt.get_lidar_coor(*cam)
        xs = vt.frustum[0, 0, :, 0].contiguous(); ys = vt.frustum[0, :, 0, 1].contiguous()
        ds = vt.frustum[:, 0, 0, 2].contiguous()
        got = E.lidar_coor(xs, ys, ds, cam)
        assert not torch.isnan(got).any()
        assert (got - exp).abs().max().item() < 2e-4   # metres; closed-form vs LU inverse


from da_cases import da_case as _da_case  # noqa: E402  (shared with the GPU test of the fused backward)


def _interleave(v):
    """(..., M, HS) head-major token rows -> the same floats stored (HS/4, M, 4), shape label kept"""
    M, HS = v.shape[-2:]
    return v.reshape(v.shape[:-2] + (M, HS // 4, 4)).transpose(-3, -2).contiguous().view(v.shape)


def _deinterleave(v):
    M, HS = v.shape[-2:]
    return v.reshape(v.shape[:-2] + (HS // 4, M, 4)).transpose(-3, -2).contiguous().view(v.shape)


def test_fused_da_cross_attention_emulated():
    for seed, kw in ((0, {}), (1, dict(B=1, Q=33, shapes=((4, 6),), P=4, M=2, E=8)),
                     (2, dict(B=1, Q=41, E=40, M=4)),                      # Dh = 10: unit-per-lane kernel (FB-OCC)
                     (3, dict(B=2, Q=19, E=16, M=2, shapes=((6, 5), (3, 3), (2, 2)), P=4))):   # Dh = 8, 3 levels
        args, exp = _da_case(seed, **kw)
        got = E.da_cross_attn_fwd(*args)
        assert not torch.isnan(got).any()
        assert torch.allclose(got, exp, atol=2e-5, rtol=1e-5)
        if args[0].shape[-1] in (8, 10, 16, 32):
            # both kernels evaluate the same expressions in the same order: identical bits
            assert torch.equal(got, E.da_cross_attn_fwd(*args, misalign=True))
        # head-minor layout of the two per-query tensors: (B,Q,M,L,P[,2]) -> (B,Q,L,P,M[,2]), same values => same bits
        a = list(args)
        a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
        a[8] = args[8].permute(0, 1, 3, 4, 2).contiguous()
        assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=3))
        assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=3, misalign=True))
        a[8] = args[8]                                                    # offsets head-minor only (the module's choice)
        assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=1))
        # head-padded value rows (Dh -> multiple of 4, 16-byte aligned head chunks): same bits, padding ignored
        Dh = args[0].shape[-1]
        HS = (Dh + 3) // 4 * 4 + (4 if Dh % 4 == 0 else 0)
        vp = torch.full(args[0].shape[:-1] + (HS,), 7.0e5)              # garbage in the padding must not matter
        vp[..., :Dh] = args[0]
        a[0] = vp
        assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=1, head_dim=Dh))
        assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=1, head_dim=Dh, misalign=True))
        # chunk-major token rows (HS/4, M, 4) -- head_minor bit 2: same floats at other addresses => same bits
        a[0] = _interleave(vp)
        for hm in (4, 5):
            a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous() if hm & 1 else args[7]
            assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=hm, head_dim=Dh))
            assert torch.equal(got, E.da_cross_attn_fwd(*a, head_minor=hm, head_dim=Dh, misalign=True))
        # 16-bit tokens (fbbev_da_cross_attn_fwd_e): rows chunk-major with 8-element pieces; the elements are widened
        # exactly, so the result is the fp32 kernel's on the rounded tokens, bit for bit
        if Dh in (8, 10, 16, 32):
            HS16 = (Dh + 7) // 8 * 8
            for dt in (torch.bfloat16, torch.float16):
                v16 = torch.full(args[0].shape[:-1] + (HS16,), 3.0e4).to(dt)
                v16[..., :Dh] = args[0].to(dt)
                M_ = v16.shape[-2]
                il = v16.reshape(v16.shape[:-2] + (M_, HS16 // 8, 8)).transpose(-3, -2).contiguous().view(v16.shape)
                a2 = list(args)
                a2[0] = args[0].to(dt).float()
                want = E.da_cross_attn_fwd(*a2)
                a2[0] = il
                a2[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
                assert torch.equal(E.da_cross_attn_fwd(*a2, head_minor=5, head_dim=Dh), want), dt


def test_pipelined_da_cross_attention_emulated():
    """fbbev_da_cross_attn_fwd_zt -> k_da_cross_attn_fwd_pipe (two samples in flight per lane; padded corners and
    out-of-image samples read the zero token behind the value rows): against the oracle's composite result and against
    the one-sample-at-a-time unit kernel on the same inputs.  Not bit-identical by design -- `offset / size` is evaluated as
    offset * (1 / size) -- but within fp32 rounding of a continuous function.  A zero-token filled with a NON-zero value
    must change the result exactly when some sample has a padded corner (it does in every case below): proof that the
    token is what those corners read; shapes outside the pipelined kernel's preconditions fall back to the unit kernel."""
    cases = ((2, dict(B=1, Q=41, E=40, M=4)),                                          # Dh = 10, 2 levels x 8 points (LP = 16)
             (4, dict(B=2, Q=300, E=80, M=8, shapes=((16, 44),), P=8, DC=20)),          # the shipped head layout, > 1 workgroup
             (3, dict(B=2, Q=19, E=16, M=2, shapes=((6, 5), (3, 3), (2, 2)), P=4)),      # Dh = 8, 3 levels x 4 points
             (6, dict(B=1, Q=37, E=40, M=4, shapes=((5, 7), (9, 6), (3, 4), (2, 2)), P=8)))   # 4 levels: LP = 32
    for seed, kw in cases:
        args, exp = _da_case(seed, **kw)
        Dh = args[0].shape[-1]
        HS = (Dh + 3) // 4 * 4
        vp = torch.zeros(args[0].shape[:-1] + (HS,))
        vp[..., :Dh] = args[0]
        a = list(args)
        a[0] = _interleave(vp)
        a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
        unit = E.da_cross_attn_fwd(*a, head_minor=5, head_dim=Dh)
        pipe = E.da_cross_attn_fwd(*a, head_minor=5, head_dim=Dh, zero_token=0.0)
        assert not torch.isnan(pipe).any()
        assert torch.allclose(pipe, exp, atol=2e-5, rtol=1e-5), (seed, (pipe - exp).abs().max())
        assert torch.allclose(pipe, unit, atol=2e-6, rtol=1e-5), (seed, (pipe - unit).abs().max())
        poisoned = E.da_cross_attn_fwd(*a, head_minor=5, head_dim=Dh, zero_token=3.0)
        assert not torch.equal(poisoned, pipe), seed                                    # the token IS read ...
        assert torch.isfinite(poisoned).all()
        # FBBEV_DA_ATTN_LOGITS (0x10): raw attention logits in, softmax over each unit's L*P weights fused into the LDS staging
        lg = list(a)
        lg[8] = (args[8].flatten(-2).log() + torch.randn(args[8].shape[:3] + (1,), generator=torch.Generator().manual_seed(seed)) * 3
                 ).view(args[8].shape).contiguous()                                     # softmax(log p + c) == p
        fused = E.da_cross_attn_fwd(*lg, head_minor=5 | 0x10, head_dim=Dh, zero_token=0.0)
        assert torch.allclose(fused, pipe, atol=2e-6, rtol=2e-5), (seed, (fused - pipe).abs().max())
    # patch mapping (bev_w given, M = 8): a workgroup = the 8 heads of an 8 x 4 patch of the BEV grid -- the SAME bits as the
    # linear unit order, for grids that are / are not multiples of the patch (partial patches masked), with fused softmax
    for seed, (bh, bw) in ((4, (12, 25)), (5, (8, 16)), (6, (5, 7))):
        args, exp = _da_case(seed, B=2, Q=bh * bw, E=80, M=8, shapes=((16, 44), (8, 22)), P=8, DC=20)
        vp = torch.zeros(args[0].shape[:-1] + (12,)); vp[..., :10] = args[0]
        a = list(args); a[0] = _interleave(vp); a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
        lin = E.da_cross_attn_fwd(*a, head_minor=5, head_dim=10, zero_token=0.0)
        pat = E.da_cross_attn_fwd(*a, head_minor=5, head_dim=10, zero_token=0.0, bev_w=bw)
        assert torch.equal(lin, pat), (seed, (lin - pat).abs().max())
        assert torch.allclose(pat, exp, atol=2e-5, rtol=1e-5)
        lg = list(a); lg[8] = args[8].flatten(-2).log().view(args[8].shape).contiguous()
        assert torch.equal(E.da_cross_attn_fwd(*lg, head_minor=5 | 0x10, head_dim=10, zero_token=0.0),
                           E.da_cross_attn_fwd(*lg, head_minor=5 | 0x10, head_dim=10, zero_token=0.0, bev_w=bw))
    # ... and outside the preconditions (here Za = 2; head-major rows) the entry runs the unit kernel: identical bits
    args, exp = _da_case(7, B=1, Q=23, E=40, M=4, Za=2)
    Dh = args[0].shape[-1]
    vp = torch.zeros(args[0].shape[:-1] + (12,)); vp[..., :Dh] = args[0]
    a = list(args); a[0] = _interleave(vp); a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
    assert torch.equal(E.da_cross_attn_fwd(*a, head_minor=5, head_dim=Dh, zero_token=3.0),
                       E.da_cross_attn_fwd(*a, head_minor=5, head_dim=Dh))
    args, exp = _da_case(2, B=1, Q=41, E=40, M=4)
    vp = torch.zeros(args[0].shape[:-1] + (12,)); vp[..., :10] = args[0]
    a = list(args); a[0] = vp; a[7] = args[7].permute(0, 1, 3, 4, 2, 5).contiguous()
    assert torch.equal(E.da_cross_attn_fwd(*a, head_minor=1, head_dim=10, zero_token=3.0),
                       E.da_cross_attn_fwd(*a, head_minor=1, head_dim=10))


def test_tokens_from_nchw_emulated():
    """fbbev_tokens_from_nchw: per-level transposition + cams_embeds into the (bs*num_cam, sum HW, C) token rows ==
    bevformer.py:95-117 (flatten/permute/add, cat) followed by the rebatch permute; and the plain inverse transposition."""
    g = torch.Generator().manual_seed(3)
    bs, ncam, C = 2, 3, 40
    shapes = [(5, 7), (10, 13), (2, 3)]
    feats = [torch.randn(bs, ncam, C, h, w, generator=g) for h, w in shapes]
    ce = torch.randn(ncam, C, generator=g)
    S_ = sum(h * w for h, w in shapes)
    rows = torch.full((bs * ncam, S_, C), float('nan'))
    start = 0
    for f, (h, w) in zip(feats, shapes):
        E.tokens_from_nchw(f.reshape(bs * ncam, C, h * w).contiguous(), rows, start * C, ce)
        start += h * w
    ref = torch.cat([f.flatten(3).permute(1, 0, 3, 2) + ce[:, None, None, :] for f in feats], 2)     # (ncam, bs, S, C)
    ref = ref.permute(0, 2, 1, 3).permute(2, 0, 1, 3).reshape(bs * ncam, S_, C)
    assert torch.equal(rows, ref)
    x = torch.randn(2, 37, 50, generator=g)
    out = E.tokens_from_nchw(x, torch.full((2, 50, 37), float('nan')))
    assert torch.equal(out, x.transpose(1, 2).contiguous())
    # per-position row added in the same pass (the BEV queries: lss_bev tokens + bev_embedding)
    pos = torch.randn(50, 37, generator=g)
    out = E.tokens_from_nchw(x, torch.full((2, 50, 37), float('nan')), pos_bias=pos)
    assert torch.equal(out, x.transpose(1, 2) + pos[None])


def test_point_sampling_emulated():
    cfg = S.CONFIGS['REF']
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
    gcb = {'x': [-40, 40, 4.0], 'y': [-40, 40, 4.0], 'z': [-1, 5.4, 1.6]}          # 20x20x4 voxel centres
    ref3d = O.reference_points_3d(gcb)
    exp_ref, exp_mask, exp_d = O.point_sampling(ref3d, cam, (256, 704), inverse=O.inv3x3_closed_form)
    xs = ref3d[0, :, 0, 0].contiguous(); ys = ref3d[:, 0, 0, 1].contiguous(); zs = ref3d[0, 0, :, 2].contiguous()
    ref, mask, qd = E.point_sampling(xs, ys, zs, cam, 256, 704)
    assert not torch.isnan(ref).any() and not torch.isnan(qd).any()
    assert (mask != exp_mask).sum().item() <= 2            # borderline points may flip (different fp32 op order)
    same = (mask == exp_mask)
    vis = exp_mask & same
    assert torch.allclose(ref[vis], exp_ref[vis], atol=2e-5)
    assert torch.allclose(qd[vis], exp_d.squeeze(-1)[vis], atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize('tv,flags', [(64, 0x100000), (128, 0x100004), (256, 0x104405), (1024, 0x100001), (16, 0x120000),
                                      (8, 0x100400), (32, 0x102401)])
def test_pool_dense_channels_last_matches_oracle(tv, flags):
    """(B,Z,Y,X,C) dense output == the reference op's layout with every row written once."""
    for name, B in (('TINY', 2), ('SMALL', 1)):
        cfg, vt, coor, depth, feat = _case(name, B)
        rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
        Bz, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
        code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
        assert code == 0 and not torch.isnan(out).any()
        erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
        exp = O.bev_pool_v2_fwd(depth, feat, erd, erf, erb, (B, Z, Y, X, C), est, eln, use_fma=True)
        assert torch.equal(out, exp)


@pytest.mark.parametrize('name,B', [('TINY', 2), ('SMALL', 1)])
def test_fused_geometry_rank_build_equals_two_step(name, B):
    """fbbev_lift_rank_build (keys evaluated inside the sort's first pass) == fbbev_lidar_coor + fbbev_rank_build."""
    cfg = S.CONFIGS[name]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=0, bda_aug=True)
    xs = vt.frustum[0, 0, :, 0].contiguous(); ys = vt.frustum[0, :, 0, 1].contiguous(); ds = vt.frustum[:, 0, 0, 2].contiguous()
    coor = E.lidar_coor(xs, ys, ds, cam)
    two = E.rank_build(coor, *_grid3(vt))
    P, I = two[6].tolist()
    for frustum in (None, vt.frustum.contiguous()):
        one = E.lift_rank_build(xs, ys, ds, cam, *_grid3(vt), frustum=frustum)
        assert one[6].tolist() == [P, I] and P > 0
        for a, b, n in zip(one[:6], two[:6], (P, P, P, I, I, I)):
            assert torch.equal(a[:n], b[:n])


@pytest.mark.parametrize('name,B,tv,flags', [('TINY', 2, 64, 0), ('SMALL', 1, 128, 0x24424), ('TINY', 2, 64, 0x800414)])
def test_lift_splat_fused_one_entry_equals_the_four_calls_emulated(name, B, tv, flags):
    """fbbev_lift_splat_fused (view_transformer.py:521-545 as ONE C entry) == fbbev_lift_rank_build -> fbbev_nchw_to_nhwc ->
    fbbev_pool_tile_index -> fbbev_bev_pool_v2_dense_fwd bit for bit, == the C oracle on the same coor; the index tensors it
    leaves in its workspace are the ones the four calls produce; the camera-keyed form keeps them on a hit and rebuilds on a
    changed rig; bf16 storage; short / misaligned workspace refused."""
    import ctypes
    cfg = S.CONFIGS[name]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=0, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, B, seed=0)
    xs = vt.frustum[0, 0, :, 0].contiguous(); ys = vt.frustum[0, :, 0, 1].contiguous(); ds = vt.frustum[:, 0, 0, 2].contiguous()
    _, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
    N, D, (H, W) = cfg.n_cams, cfg.D, cfg.feat_hw
    L = E.lib()
    rb, rd, rf, st, ln, ir, counts = E.lift_rank_build(xs, ys, ds, cam, *_grid3(vt))
    feat = E.nchw_to_nhwc(ctx)
    code, four = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags & ~0x800000)
    assert code == 0
    exp = O.bev_pool_v2(depth, feat, *[t[:counts[0]] for t in (rd, rf, rb)], (B, Z, Y, X, C), st[:counts[1]], ln[:counts[1]], use_fma=True)
    assert torch.equal(four, exp)
    dims = (B, N, D, H, W, C, Z, Y, X)
    need = L.fbbev_lift_splat_fused_ws_bytes(*dims)
    assert need > 0 and L.fbbev_lift_splat_fused_ws_bytes(0, N, D, H, W, C, Z, Y, X) == 0
    ws = torch.full((need,), 0xA5, dtype=torch.uint8)
    arr = ctypes.c_float * 3
    lo, it, gs = (arr(*v) for v in _grid3(vt))
    bf16 = bool(flags & 0x800000)
    out = torch.full((B, C, Z, Y, X), float('nan'), dtype=torch.bfloat16 if bf16 else torch.float32)

    def call(cam_, ws_, key=None, state=None, nbytes=None):
        rots, trans, intrins, post_rots, post_trans, bda = cam_
        return L.fbbev_lift_splat_fused(None, E.p(xs), E.p(ys), E.p(ds), E.p(rots), E.p(trans), E.p(intrins), E.p(post_rots), E.p(post_trans),
                                        E.p(bda), E.p(depth), E.p(ctx), B, N, D, H, W, C, ctypes.cast(lo, ctypes.c_void_p),
                                        ctypes.cast(it, ctypes.c_void_p), ctypes.cast(gs, ctypes.c_void_p), Z, Y, X, E.p(out), 0, 0, tv, flags,
                                        ctypes.c_void_p(ws_.data_ptr()), ws_.numel() if nbytes is None else nbytes,
                                        None if key is None else E.p(key), None if state is None else E.p(state), None)
    assert call(cam, ws) == 0 and not torch.isnan(out.float()).any()
    assert torch.equal(out, four.to(out.dtype))                           # 16-bit storage = the fp32 sums rounded once
    off = (ctypes.c_size_t * 8)()
    assert L.fbbev_lift_splat_fused_ws_offsets(*dims, ctypes.cast(off, ctypes.c_void_p)) == 0
    P, I = ws[off[6]:off[6] + 8].view(torch.int32).tolist()
    assert [P, I] == counts.tolist()
    for k, (ref, n) in enumerate(((rb, P), (rd, P), (rf, P), (st, I), (ln, I), (ir, I))):
        assert torch.equal(ws[off[k]:off[k] + 4 * n].view(torch.int32), ref[:n]), k
    assert torch.equal(ws[off[7]:off[7] + 4 * feat.numel()].view(torch.float32), feat.reshape(-1))
    # camera-keyed cache: first call builds, second (same rig) keeps the workspace's index tensors, a changed rig rebuilds
    key = torch.full((L.fbbev_cam_key_words(B, N),), -1, dtype=torch.int32)
    state = torch.tensor([0, 0, -1, -1], dtype=torch.int32)
    ws2 = torch.full((need,), 0x5A, dtype=torch.uint8)
    for want_hit, want_builds, cam_ in ((0, 1, cam), (1, 1, cam), (0, 2, S.camera_rig(cfg, B, seed=3, bda_aug=True)), (0, 3, cam)):
        out.fill_(float('nan'))
        assert call(cam_, ws2, key, state) == 0
        assert state[:2].tolist() == [want_hit, want_builds] and not torch.isnan(out.float()).any()
        if cam_ is cam:
            assert torch.equal(out, four.to(out.dtype))
    # refused: workspace too small / one of the cache pointers missing
    assert call(cam, ws, nbytes=need - 256) == -4 or call(cam, ws, nbytes=need - 256) < 0
    assert call(cam, ws2, key, None) < 0


def test_nchw_to_nhwc_emulated():
    for shape in ((2, 3, 8, 4, 6), (1, 2, 80, 5, 7), (1, 1, 33, 3, 11)):
        x = torch.randn(shape, generator=torch.Generator().manual_seed(0))
        assert torch.equal(E.nchw_to_nhwc(x), x.permute(0, 1, 3, 4, 2).contiguous())


def _bwd_expected(og_ncdhw, depth, feat, rb, rd, rf):
    """Oracle backward with the points of a pixel taken in ascending point id (= ascending depth bin)."""
    o = torch.argsort(rd.long())
    return O.bev_pool_v2_bwd(og_ncdhw.permute(0, 2, 3, 4, 1).contiguous(), depth, feat, rd[o].contiguous(),
                             rf[o].contiguous(), rb[o].contiguous())


# D=40 (two 32-bin chunks per pixel) and C=136 (> 128: the 8-channels-per-lane instantiation)
DEEP = S.PathConfig('DEEP', (32, 48), 8, {'x': [-8, 8, 1.0], 'y': [-8, 8, 1.0], 'z': [-1, 3, 1.0],
                                           'depth': [1.0, 9.0, 0.2]}, 136, n_cams=2)


@pytest.mark.parametrize('name,B,padded', [('TINY', 2, False), ('TINY', 1, True), ('SMALL', 1, False),
                                           (DEEP, 1, False)])
def test_pool_dense_bwd_emulated(name, B, padded):
    cfg, vt, coor, depth, feat = _case(name, B)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    _, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
    g = torch.Generator().manual_seed(5)
    if padded:      # gradient living inside a wider (B, C+4, Z, Y, X) buffer: strides are honoured
        og = torch.randn((B, C + 4, Z, Y, X), generator=g)[:, 2:2 + C]
    else:
        og = torch.randn((B, C, Z, Y, X), generator=g)
    code, dg, fg = E.pool_dense_bwd(og, depth, feat, rd, ir, st, counts, st.numel(), (Z, Y, X))
    assert code == 0
    assert not torch.isnan(dg).any() and not torch.isnan(fg).any()     # both written completely, zeros included
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    edg, efg = _bwd_expected(og, depth, feat, erb, erd, erf)
    assert torch.equal(fg, efg)                                        # in-order fmaf chain, ascending depth bin
    assert torch.allclose(dg, edg, atol=1e-5, rtol=1e-5)               # lane-tree vs serial channel sum
    kept = torch.zeros(depth.numel(), dtype=torch.bool)
    kept[erd.long()] = True
    assert torch.equal(dg.flatten()[~kept], torch.zeros((~kept).sum()))
    # fbbev_bev_pool_v2_dense_bwd_z: a (B,C,Y,X) gradient added to every z plane inside the gradient read (the Z-mean's backward)
    # == the plain backward of out_grad + zscale * zgrad[:, :, None], bit for bit
    zg = torch.randn((B, C, Y, X), generator=g)
    code, dg2, fg2 = E.pool_dense_bwd(og, depth, feat, rd, ir, st, counts, st.numel(), (Z, Y, X), zgrad=zg, zscale=1.0 / Z)
    code3, dg3, fg3 = E.pool_dense_bwd((og + (zg * (1.0 / Z))[:, :, None]).contiguous(), depth, feat, rd, ir, st, counts, st.numel(), (Z, Y, X))
    assert code == 0 and code3 == 0 and torch.equal(dg2, dg3) and torch.equal(fg2, fg3)


def test_pool_dense_bwd_empty_index_writes_zeros():
    cfg, vt, coor, depth, feat = _case('TINY', 1)
    coor = coor + 1.0e4                                                # every point leaves the grid
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    assert counts.tolist() == [0, 0]
    _, Z, Y, X, C = vt.bev_feat_shape(1, cfg.channels)
    og = torch.randn((1, C, Z, Y, X), generator=torch.Generator().manual_seed(1))
    code, dg, fg = E.pool_dense_bwd(og, depth, feat, rd, ir, st, counts, st.numel(), (Z, Y, X))
    assert code == 0 and not dg.any() and not fg.any()


# ------------------------------------------------------------------ temporal history alignment
def _history_fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'history_fusion_seq4.npz'))
    return z, [int(v) for v in z['dims']]


def test_history_flow_and_warp_emulated_vs_reference_fixture():
    """k_history_flow + k_history_warp against the grid / sampled volume recorded from the real
    FBOCC.generate_grid + F.grid_sample (fbocc.py:169-205,264-275)."""
    from oracle import history_oracle as H
    z, (B, C, T, Z, Y, X) = _history_fixture()
    dx, bx = z['dx'], z['bx']
    lower = bx - dx / 2
    hist_augs = None
    for i in range(4):
        f = {k: torch.from_numpy(z[f'f{i}.{k}']) for k in ('curr', 'bda', 'ego', 'start', 'grid', 'sampled', 'history_after')}
        fwd = H.forward_aug_matrix(f['bda'])
        if hist_augs is None:
            hist_augs = fwd.clone()
        hist_augs[f['start'].bool()] = fwd[f['start'].bool()]
        flow = E.history_flow(hist_augs.contiguous(), f['ego'].contiguous(), f['bda'].contiguous(), dx, lower)
        eflow = H.rt_flow(hist_augs, fwd, f['ego'], torch.from_numpy(dx), torch.from_numpy(bx))
        assert torch.allclose(flow, eflow, atol=2e-5, rtol=1e-5), i
        grid = H.generate_grid(flow, (Z, Y, X)).permute(0, 3, 1, 2, 4)
        assert torch.allclose(grid, f['grid'], atol=3e-5), i
        # history entering frame i = history_after of frame i-1 with restarted samples overwritten (fbocc.py:253-257)
        if i == 0:
            hist = f['curr'].permute(0, 1, 4, 2, 3).repeat(1, T, 1, 1, 1).contiguous()
        else:
            hist = torch.from_numpy(z[f'f{i - 1}.history_after']).clone()
            st = f['start'].bool()
            hist[st] = f['curr'].permute(0, 1, 4, 2, 3)[st].repeat(1, T, 1, 1, 1)
        out = E.history_warp(hist, flow)
        assert not torch.isnan(out).any()
        assert torch.allclose(out, f['sampled'], atol=2e-4), i                # vs the reference's own kernel
        assert torch.allclose(out, H.warp_history(hist, flow), atol=1e-5), i  # vs the oracle on the same rt_flow
        hist_augs = fwd.clone()


def test_history_warp_strided_output_and_padding_emulated():
    from oracle import history_oracle as H
    g = torch.Generator().manual_seed(4)
    B, CH, Z, Y, X = 2, 11, 3, 9, 21                                         # odd channel count: tail path
    hist = torch.randn(B, CH, Z, Y, X, generator=g)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([2.5, -1.25, 0.5])                         # translation: part of the volume leaves the grid
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    big = torch.full((B, CH + 5, Z, Y, X), float('nan'))
    E.history_warp(hist, flow, big[:, 5:])                                   # write into a channel slice (batch stride)
    exp = H.warp_history(hist, flow)
    assert torch.allclose(big[:, 5:], exp, atol=1e-5)
    assert torch.isnan(big[:, :5]).all()
    assert (exp[0, :, :, :, -2:] == 0).all()                                 # x + 2.5 > X-1: zero padding
    ident = E.history_warp(hist, torch.eye(4)[None].repeat(B, 1, 1).contiguous())
    assert torch.allclose(ident, hist, atol=1e-6)
    nanflow = flow.clone(); nanflow[0, 0, 0] = float('nan')
    assert (E.history_warp(hist, nanflow)[0] == 0).all()                     # NaN coordinates sample nothing


@pytest.mark.parametrize('tv,flags,dt', [(64, 0x800000, torch.bfloat16), (128, 0x824424, torch.bfloat16),
                                         (64, 0x1000000, torch.float16), (256, 0x1020024, torch.float16)])
def test_pool_dense_16bit_storage_is_the_rounded_fp32_result(tv, flags, dt):
    """BASELINE configs[1] (bf16) / configs[4] (fp16) storage: the fp32 in-order sums, rounded once to nearest-even
    at the store -- bit-identical to torch's own fp32 -> bf16 / fp16 conversion of the fp32 output."""
    cfg, vt, coor, depth, feat = _case('TINY', 2)                    # Y*X = 256: a multiple of 8
    depth = depth * 37.0                                             # spread the sums over more binades
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    B, Z, Y, X, C = vt.bev_feat_shape(2, cfg.channels)
    code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    assert code == 0 and out.dtype == dt
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    exp = O.bev_pool_v2(depth, feat, erd, erf, erb, (B, Z, Y, X, C), est, eln, use_fma=True)
    assert torch.equal(out.view(torch.int16), exp.to(dt).view(torch.int16))


def test_f16_conversion_edge_cases_emulated():
    """fbbev_f32_to_f16 through the kernel: subnormals, ties, overflow, NaN."""
    cfg, vt, coor, depth, feat = _case('TINY', 1)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    B, Z, Y, X, C = vt.bev_feat_shape(1, cfg.channels)
    vals = torch.tensor([6.0e-8, 5.9604645e-8 * 1.5, 6.1035e-5, 65504.0, 65520.0, 1.0e6, 1.0009765625, 1.00048828125],
                        dtype=torch.float32)
    feat = vals.view(1, 1, 1, 1, C).expand_as(feat).contiguous()     # C == 8 channels carry the probe values
    depth = torch.zeros_like(depth)
    P = int(counts[0])
    first = rd[:P].long()[st[:int(counts[1])].long()]                # first point of every interval: depth 1, rest 0
    depth.view(-1)[first] = 1.0
    for flags, dt in ((0x1000000, torch.float16), (0x800000, torch.bfloat16)):
        code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, 64, flags)
        assert code == 0
        got = out.permute(0, 2, 3, 4, 1).reshape(-1, C)
        hit = got.float().abs().sum(1) > 0
        assert hit.any()
        assert torch.equal(got[hit].view(torch.int16), vals.to(dt).view(torch.int16).expand(int(hit.sum()), C))


@pytest.mark.parametrize('rows,C', [(37, 80), (8, 128), (5, 4), (64, 64)])
def test_layernorm_rows_emulated(rows, C):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g) * 3 + 1.5
    r = torch.randn(rows, C, generator=g)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    exp = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
    assert torch.allclose(E.layernorm(x, w, b, 1e-5), exp, atol=2e-6, rtol=1e-5)
    exp2 = torch.nn.functional.layer_norm(x + r, (C,), w, b, 1e-5)
    assert torch.allclose(E.layernorm(x, w, b, 1e-5, residual=r), exp2, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize('R,I,O,relu', [(300, 80, 128, False), (130, 80, 64, False), (257, 80, 96, False), (129, 80, 512, True),
                                        (200, 512, 80, False), (64, 80, 320, True), (50, 8, 4, False), (1, 264, 132, True)])
def test_rows_linear_split_operand_emulated(R, I, O, relu):
    """fbbev_rows_linear_x3 (x W^T + b [+ ReLU], split-operand bf16 MFMA) against float64: < 2e-5 of the output peak for the
    backward projection's layer shapes (80 -> 64 / 96 / 128 / 320 / 512, 512 -> 80: four K chunks), partial row tiles, partial output
    chunks (132 = 128 + 4), an input width that is not a multiple of 32, no bias, strided rows in and out."""
    g = torch.Generator().manual_seed(R + I + O)
    xs = torch.randn(R, I + 8, generator=g) * 2
    x = xs[:, :I]                                                      # row stride I + 8
    w = torch.randn(O, I, generator=g) * 0.2
    b = torch.randn(O, generator=g) if O != 64 else None
    outs = torch.full((R, O + 4), float('nan'))
    code, got = E.rows_linear_x3(x, w, b, relu=relu, out=outs[:, :O])
    assert code == 0
    exp = x.double() @ w.double().t() + (b.double() if b is not None else 0)
    if relu:
        exp = exp.relu()
    assert not torch.isnan(got).any() and torch.isnan(outs[:, O:]).all()          # nothing written beyond the output columns
    assert (got.double() - exp).abs().max() <= 2e-5 * exp.abs().max()
    # plain bf16 operands would be ~100x further away
    xb, wb = x.bfloat16().double(), w.bfloat16().double()
    e1 = xb @ wb.t() + (b.double() if b is not None else 0)
    if relu:
        e1 = e1.relu()
    assert (got.double() - exp).abs().max() * 30 < (e1 - exp).abs().max()


def test_rows_linear_several_row_tiles_per_workgroup_emulated(monkeypatch):
    """the n_kc == 1 shape with RT = 3 row tiles per workgroup (fragments staged once): 5 tiles of 128 rows -> groups of 3 + 2,
    a partial last tile, two output chunks; same result as one tile per workgroup, bit for bit."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(5 * 128 - 37, 80, generator=g)
    w, b = torch.randn(160, 80, generator=g) * 0.2, torch.randn(160, generator=g)
    code, one = E.rows_linear_x3(x, w, b, relu=True)
    monkeypatch.setenv('FBBEV_ROWS_LINEAR_RT', '3')
    code3, three = E.rows_linear_x3(x, w, b, relu=True)
    assert code == 0 and code3 == 0 and not torch.isnan(three).any()
    assert torch.equal(one, three)


@pytest.mark.parametrize('R,I,O,relu,slots', [(5 * 128 - 37, 80, 160, True, 2), (300, 80, 80, False, 1), (130, 64, 64, True, 3),
                                              (257, 128, 96, False, 2), (129, 80, 512, True, 1), (700, 96, 80, False, 2)])
def test_rows_linear_persistent_form_equals_the_tile_per_workgroup_form_emulated(R, I, O, relu, slots, monkeypatch):
    """k_rows_linear_x3p (round 6: a workgroup stays on its output chunk and walks every n_slots-th row tile -- fragments and bias staged
    once, the next tile's rows requested under the MFMAs, compile-time tile counts) against k_rows_linear_x3 (FBBEV_ROWS_LINEAR_P=0):
    the SAME BITS, plain rows and head planes (fp32 / 16-bit), with and without bias, the training epilogue (residual / mask), partial
    last tiles, 1-3 tiles per workgroup, one to four output chunks, K of 2-4 steps."""
    g = torch.Generator().manual_seed(R + I + O)
    x = torch.randn(R, I, generator=g) * 2
    w = torch.randn(O, I, generator=g) * 0.2
    b = torch.randn(O, generator=g) if O != 64 else None
    res, msk = torch.randn(R, O, generator=g), torch.randn(R, O, generator=g)

    def run():
        outs = {}
        code, outs['rows'] = E.rows_linear_x3(x, w, b, relu=relu)
        assert code == 0
        code, outs['train'] = E.rows_linear_x3_train(x, w, b, relu=relu, residual=res, mask=msk)
        assert code == 0
        if O % 8 == 0 and R % 5 == 0 and O <= 128:
            for dt in (None, torch.bfloat16):
                code, outs[f'planes{dt}'] = E.rows_linear_x3_planes(x, w, b, R // 5, 8, O // 8, dtype=dt)
                assert code == 0
        return outs

    monkeypatch.setenv('FBBEV_ROWS_LINEAR_P', '0')
    old = run()
    monkeypatch.setenv('FBBEV_ROWS_LINEAR_P', '1')
    monkeypatch.setenv('FBBEV_ROWS_LINEAR_SLOTS', str(slots))
    new = run()
    for k in old:
        a, c = old[k], new[k]
        assert not torch.isnan(c.float()).any(), k
        assert torch.equal(a.view(torch.int16) if a.dtype != torch.float32 else a, c.view(torch.int16) if c.dtype != torch.float32 else c), k


def test_rows_linear_with_periodic_addend_emulated():
    """fbbev_rows_linear_x3_add: rows = x[r] + addend[r % P] (the query + query_pos of the attention modules folded into the
    projection), bit-identical to the plain entry on the pre-added rows; 3 samples of 150 queries, a partial last row tile."""
    g = torch.Generator().manual_seed(5)
    Q, B, I, O = 150, 3, 80, 96
    x = torch.randn(B * Q, I, generator=g)
    pos = torch.randn(Q, I, generator=g)
    w, b = torch.randn(O, I, generator=g) * 0.2, torch.randn(O, generator=g)
    code, fused = E.rows_linear_x3(x, w, b, addend=pos)
    summed = (x.view(B, Q, I) + pos[None]).reshape(B * Q, I).contiguous()
    code2, plain = E.rows_linear_x3(summed, w, b)
    assert code == 0 and code2 == 0 and not torch.isnan(fused).any()
    assert torch.equal(fused, plain)


@pytest.mark.parametrize('R,I,O', [(300, 80, 80), (1000, 80, 512), (77, 80, 96), (257, 320, 80), (64, 80, 32), (33, 8, 4), (40, 132, 260)])
def test_rows_wgrad_split_operand_emulated(R, I, O, monkeypatch):
    """fbbev_rows_wgrad_x3 (grad_weight = grad_out^T x, grad_bias = column sums: autograd's backward of nn.Linear) against float64 for
    the backward projection's layer shapes (80 x 80, 512 x 80, 96 x 80, the FFN's 80 x 320 with three input chunks), row counts that
    are not multiples of the 32-row step, widths that are not multiples of a 16-column tile, strided rows, three K splits (the last one
    partial); a second run is bit-identical, and the bias gradient can be skipped."""
    g = torch.Generator().manual_seed(R * 7 + I + O)
    gys = torch.randn(R, O + 4, generator=g)
    xs = torch.randn(R, I + 8, generator=g) * 2
    gy, x = gys[:, :O], xs[:, :I]
    monkeypatch.setenv('FBBEV_WGRAD_SPLITS', '3')
    code, gw, gb = E.rows_wgrad_x3(gy, x)
    assert code == 0 and not torch.isnan(gw).any() and not torch.isnan(gb).any()
    ew, eb = gy.double().t() @ x.double(), gy.double().sum(0)
    assert (gw.double() - ew).abs().max() <= 2e-5 * ew.abs().max()
    assert (gb.double() - eb).abs().max() <= 1e-5 * max(eb.abs().max().item(), 1.0)
    # plain bf16 operands would be far away
    e1 = gy.bfloat16().double().t() @ x.bfloat16().double()
    assert (gw.double() - ew).abs().max() * 30 < (e1 - ew).abs().max()
    code2, gw2, gb2 = E.rows_wgrad_x3(gy, x)
    assert code2 == 0 and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    code3, gw3, gb3 = E.rows_wgrad_x3(gy, x, with_bias=False)
    assert code3 == 0 and gb3 is None and torch.equal(gw, gw3)


def test_rows_wgrad_with_periodic_addend_emulated():
    """x_addend: the layer's input was x[r] + addend[r % P] -- equal to the plain entry on the pre-added rows, bit for bit"""
    g = torch.Generator().manual_seed(3)
    Q, B, I, O = 75, 3, 80, 64
    x, pos, gy = torch.randn(B * Q, I, generator=g), torch.randn(Q, I, generator=g), torch.randn(B * Q, O, generator=g)
    c1, w1, b1 = E.rows_wgrad_x3(gy, x, addend=pos)
    c2, w2, b2 = E.rows_wgrad_x3(gy, (x.view(B, Q, I) + pos[None]).reshape(B * Q, I).contiguous())
    assert c1 == 0 and c2 == 0 and torch.equal(w1, w2) and torch.equal(b1, b2)


def test_rows_linear_training_epilogue_emulated():
    """fbbev_rows_linear_x3_train: ((x [+ addend]) W^T + b) [ReLU]) * [mask > 0] + residual against the plain entry + the torch
    expressions (bit for bit: the same accumulators, one fp32 add), in place on the residual, partial tiles, two output chunks."""
    g = torch.Generator().manual_seed(8)
    R, I, O = 150, 80, 160
    x, w, b = torch.randn(R, I, generator=g), torch.randn(O, I, generator=g) * 0.2, torch.randn(O, generator=g)
    res, mask = torch.randn(R, O, generator=g), torch.randn(R, O, generator=g)
    c0, plain = E.rows_linear_x3(x, w, b)
    assert c0 == 0
    c1, got = E.rows_linear_x3_train(x, w, b, residual=res, mask=mask)
    assert c1 == 0 and torch.equal(got, torch.where(mask > 0, plain, torch.zeros(())) + res)
    acc = res.clone()
    c2, got2 = E.rows_linear_x3_train(x, w, b, residual=acc, out=acc)          # running sum, in place
    assert c2 == 0 and got2 is acc and torch.equal(acc, plain + res)
    c3, got3 = E.rows_linear_x3_train(x, w, None, relu=True)
    c4, plain3 = E.rows_linear_x3(x, w, None, relu=True)
    assert c3 == 0 and c4 == 0 and torch.equal(got3, plain3)
    pos = torch.randn(50, I, generator=g)
    c5, got5 = E.rows_linear_x3_train(x, w, b, addend=pos, residual=res)
    c6, plain5 = E.rows_linear_x3(x, w, b, addend=pos)
    assert c5 == 0 and c6 == 0 and torch.equal(got5, plain5 + res)


def test_sum_leading_emulated():
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(3, 50, 8, generator=g), torch.randn(3, 50, 8, generator=g)
    c, s = E.sum_leading(x)
    assert c == 0 and torch.equal(s, x[0] + x[1] + x[2])
    c, s2 = E.sum_leading(x, y)
    assert c == 0 and torch.equal(s2, ((x[0] + y[0]) + x[1] + y[1]) + x[2] + y[2])


def test_sum_partials_emulated():
    g = torch.Generator().manual_seed(6)
    part = torch.randn(100, 2, 20, generator=g)
    c, s = E.sum_partials(part)
    assert c == 0 and (s.view(2, 20).double() - part.double().sum(0)).abs().max() < 1e-5
    c2, s2 = E.sum_partials(part)
    assert torch.equal(s, s2)


@pytest.mark.parametrize('group', [4, 8, 16, 32])
def test_softmax_groups_forward_and_backward_emulated(group):
    g = torch.Generator().manual_seed(group)
    x = torch.randn(37, 8, group, generator=g) * 3
    gy = torch.randn(37, 8, group, generator=g)
    c, y = E.softmax_groups(x, group)
    exp = x.double().softmax(-1)
    assert c == 0 and (y.double() - exp).abs().max() < 3e-7
    c, gx = E.softmax_groups_bwd(y, gy, group)
    yd = y.double()
    ex = yd * (gy.double() - (yd * gy.double()).sum(-1, keepdim=True))
    assert c == 0 and (gx.double() - ex).abs().max() < 1e-6
    assert E.lib().fbbev_softmax_groups(E.p(x), 10, 12, E.p(y), None) == -2


def test_rows_wgrad_rejects_unsupported_shapes():
    assert E.lib().fbbev_rows_wgrad_x3_ws_bytes(100, 6, 8) == 0            # in_features % 4 != 0
    assert E.lib().fbbev_rows_wgrad_x3_ws_bytes(100, 8, 6) == 0
    assert E.lib().fbbev_rows_wgrad_x3_ws_bytes(100, 80, 80) > 0


def test_rows_linear_split_operand_rejects_unsupported_shapes():
    x = torch.randn(10, 12); w = torch.randn(8, 12)
    assert E.rows_linear_x3(x, w, None)[0] == -2                       # in_features % 8 != 0
    x = torch.randn(10, 16); w = torch.randn(6, 16)
    assert E.rows_linear_x3(x, w, None)[0] == -2                       # out_features % 4 != 0


@pytest.mark.parametrize('rows,C', [(37, 80), (8, 128), (5, 4), (3000, 80), (20000, 64)])
def test_layernorm_rows_backward_emulated(rows, C):
    """fbbev_layernorm_bwd == autograd of torch's layer_norm: input gradient per row, weight / bias gradients from the summed
    per-workgroup partial rows (20 000 rows: more rows than row slots, the grid-stride walk; 3 000: a partly filled last round)."""
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, C, generator=g) * 3 + 1.5).requires_grad_()
    w = torch.randn(C, generator=g).requires_grad_()
    b = torch.randn(C, generator=g).requires_grad_()
    gy = torch.randn(rows, C, generator=g)
    torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5).backward(gy)
    gx, gw, gb = E.layernorm_bwd(x.detach(), gy, w.detach(), 1e-5)
    assert torch.allclose(gx, x.grad, atol=3e-6, rtol=2e-5)
    assert (gw - w.grad).abs().max() <= 2e-5 * w.grad.abs().max().clamp_min(1.0)
    assert (gb - b.grad).abs().max() <= 2e-5 * b.grad.abs().max().clamp_min(1.0)


def test_fused_da_cross_attention_backward_emulated():
    """fbbev_da_cross_attn_bwd: the four gradients, pushed back to the leaves with torch autograd, against the
    double-precision autograd of the oracle's loop-for-loop restatement of the reference's training path."""
    for seed, kw in ((5, dict(B=1, Q=29, E=16, M=4)),                       # Dh = 4
                     (6, dict(B=2, Q=17, E=40, M=4, shapes=((4, 6), (2, 3))))):   # Dh = 10
        args, exp, leaves = _da_case(seed, grad=True, **kw)
        g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        wrt = [leaves['key'], leaves['pred']] + [leaves['Pm'][k] for k in sorted(leaves['Pm']) if 'output_proj' not in k]
        ref = torch.autograd.grad(exp, wrt, grad_outputs=g, retain_graph=True, allow_unused=True)
        value, ss, ls, pred4, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
        got = E.da_cross_attn_fwd(f32(value), ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), f32(offsets), f32(attn),
                                  d0, dstep)
        assert torch.allclose(got.double(), exp.detach(), atol=2e-5, rtol=1e-5)
        for hm in (0, 1, 3):
            o_in = f32(offsets).permute(0, 1, 3, 4, 2, 5).contiguous() if hm & 1 else f32(offsets)
            a_in = f32(attn).permute(0, 1, 3, 4, 2).contiguous() if hm & 2 else f32(attn)
            gv, gd, go, ga = E.da_cross_attn_bwd(f32(value), ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), o_in, a_in,
                                                 d0, dstep, f32(g), head_minor=hm)
            if hm & 1:
                go = go.permute(0, 1, 4, 2, 3, 5)
            if hm & 2:
                ga = ga.permute(0, 1, 4, 2, 3)
            mine = torch.autograd.grad([value, pred4, offsets, attn], wrt,
                                       grad_outputs=[gv.double(), gd.double(), go.double(), ga.double()],
                                       retain_graph=True, allow_unused=True)
            for a, b in zip(mine, ref):
                if b is None:
                    assert a is None or not a.any()
                    continue
                assert torch.allclose(a, b, atol=5e-5 * max(1.0, b.abs().max().item()), rtol=1e-4), hm
        # head-padded value: gradients land in the first Dh floats of every head chunk, the padding stays zero
        Dh = value.shape[-1]
        HS = (Dh + 3) // 4 * 4
        if HS != Dh:
            vp = torch.zeros(value.shape[:-1] + (HS,))
            vp[..., :Dh] = f32(value)
            gv2, gd2, go2, ga2 = E.da_cross_attn_bwd(vp, ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), f32(offsets), f32(attn),
                                                     d0, dstep, f32(g), head_minor=0, head_dim=Dh)
            gv0, gd0, go0, ga0 = E.da_cross_attn_bwd(f32(value), ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), f32(offsets),
                                                     f32(attn), d0, dstep, f32(g), head_minor=0)
            assert torch.equal(gv2[..., :Dh], gv0) and not gv2[..., Dh:].any()
            assert torch.equal(gd2, gd0) and torch.equal(go2, go0) and torch.equal(ga2, ga0)
        # chunk-major token rows (head_minor bit 2): the value gradient comes back in the same storage order
        vp = torch.zeros(value.shape[:-1] + (HS,))
        vp[..., :Dh] = f32(value)
        gv0, gd0, go0, ga0 = E.da_cross_attn_bwd(vp, ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), f32(offsets), f32(attn),
                                                 d0, dstep, f32(g), head_minor=0, head_dim=Dh)
        gv3, gd3, go3, ga3 = E.da_cross_attn_bwd(_interleave(vp), ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth),
                                                 f32(offsets), f32(attn), d0, dstep, f32(g), head_minor=4, head_dim=Dh)
        # (atomics: the emulator runs lanes in a fixed order, so even the value gradient is reproducible)
        assert torch.allclose(_deinterleave(gv3), gv0, rtol=1e-5, atol=1e-6)
        assert torch.equal(gd3, gd0) and torch.equal(go3, go0) and torch.equal(ga3, ga0)
        # value gradient through LDS planes + partial buffer (fbbev_da_cross_attn_bwd_ws): same sums in another order;
        # several query chunks per (sample, head) so that the reduction over chunks is exercised
        import os
        shapes_host = [tuple(int(x) for x in hw) for hw in ss.tolist()]
        # two routes: output-owned planes (the default since round 4: hit lists, one launch over all regions, no partials) and the
        # chunked scatter (FBBEV_DA_BWD_OWNED=0: query chunks, partial planes, a reduction)
        for chunks, threads, tokens, prepass in (('own', '256', None, None), ('own', '512', None, None), ('own', '256', '8', None),
                                                 ('own1', '256', '8', None),     # owned planes, bands of rows, a single copy
                                                 ('1', '256', None, None), ('3', '256', None, None), ('2', '512', None, None),
                                                 ('2', '256', '8', '0'),         # token regions: bands of rows, levels apart
                                                 ('2', '256', '8', '1')):        # + the per-(camera, query) pre-pass (the default)
            owned = chunks.startswith('own')
            os.environ['FBBEV_DA_BWD_OWNED'] = '1' if owned else '0'
            if chunks == 'own1':
                os.environ['FBBEV_DA_BWD_COPIES'] = '1'
            os.environ['FBBEV_DA_BWD_CHUNKS'] = '1' if owned else chunks
            os.environ['FBBEV_DA_BWD_THREADS'] = threads
            if tokens:
                os.environ['FBBEV_DA_BWD_TOKENS'] = tokens
            if prepass == '0':
                os.environ['FBBEV_DA_BWD_PREPASS'] = '0'
            if chunks == 'own':          # the owned route is planned (its workspace = table + hit lists, not partial planes)
                import ctypes
                flat = [int(x) for hw in shapes_host for x in hw]
                harr = (ctypes.c_int32 * len(flat))(*flat)
                Ncam_, B_, Q_, _ = mask.shape
                sizes = []
                for flag in ('1', '0'):
                    os.environ['FBBEV_DA_BWD_OWNED'] = flag
                    sizes.append(E.lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B_, Ncam_, vp.shape[1], vp.shape[2], Dh, Q_, HS,
                                                                             len(shapes_host), attn.shape[-1], mask.shape[3], harr))
                os.environ['FBBEV_DA_BWD_OWNED'] = '1'
                records = B_ * Ncam_ * Q_ * 16 * 4                       # 64-byte hit records in list order
                assert records <= sizes[0] <= records + 4 * 256 + B_ * Ncam_ * vp.shape[1] * vp.shape[2] * Dh * 4 + 256 and sizes[1] > 0 and sizes[0] != sizes[1]
                # the query without the anchor count covers BOTH routes (a launch whose owned plan fails must still find
                # room for the chunked planes); more anchors than a hit record holds -> the chunked route's size
                both = E.lib().fbbev_da_cross_attn_bwd_ws_bytes(B_, Ncam_, vp.shape[1], vp.shape[2], Dh, Q_, HS, len(shapes_host), attn.shape[-1], harr)
                assert both == max(sizes)
                assert E.lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B_, Ncam_, vp.shape[1], vp.shape[2], Dh, Q_, HS, len(shapes_host),
                                                                   attn.shape[-1], 8, harr) == sizes[1]
            if prepass == '1':
                import ctypes
                flat = [int(x) for hw in shapes_host for x in hw]
                harr = (ctypes.c_int32 * len(flat))(*flat)
                Ncam_, B_, Q_, _ = mask.shape
                wsb = lambda: E.lib().fbbev_da_cross_attn_bwd_ws_bytes(B_, Ncam_, vp.shape[1], vp.shape[2], Dh, Q_, HS, len(shapes_host),  # noqa: E731
                                                                       attn.shape[-1], harr)
                os.environ['FBBEV_DA_BWD_PREPASS'] = '0'
                plain = wsb()
                os.environ['FBBEV_DA_BWD_PREPASS'] = prepass
                assert wsb() > plain > 0                                 # the pre-pass is planned (its table sits behind the partials)
            try:
                for hm, vin in ((0, vp), (4, _interleave(vp)), (5, _interleave(vp))):
                    o_in = f32(offsets).permute(0, 1, 3, 4, 2, 5).contiguous() if hm & 1 else f32(offsets)
                    gv4, gd4, go4, ga4 = E.da_cross_attn_bwd(vin, ss, ls, f32(pred4), f32(ref_cam), mask, f32(qdepth), o_in,
                                                             f32(attn), d0, dstep, f32(g), head_minor=hm, head_dim=Dh,
                                                             lds_planes=True, level_hw=shapes_host)
                    if hm & 1:
                        go4 = go4.permute(0, 1, 4, 2, 3, 5)
                    gv4 = _deinterleave(gv4) if hm & 4 else gv4
                    assert not torch.isnan(gv4).any()
                    assert torch.allclose(gv4, gv0, rtol=1e-5, atol=1e-6) and not gv4[..., Dh:].any()
                    assert torch.allclose(gd4, gd0, rtol=1e-5, atol=1e-6)
                    assert torch.allclose(go4, go0, rtol=1e-5, atol=1e-6) and torch.allclose(ga4, ga0, rtol=1e-5, atol=1e-6)
            finally:
                del os.environ['FBBEV_DA_BWD_CHUNKS'], os.environ['FBBEV_DA_BWD_THREADS']
                os.environ.pop('FBBEV_DA_BWD_TOKENS', None)
                os.environ.pop('FBBEV_DA_BWD_PREPASS', None)
                os.environ.pop('FBBEV_DA_BWD_OWNED', None)
                os.environ.pop('FBBEV_DA_BWD_COPIES', None)


def test_da_backward_unit_gradients_on_head_planes_emulated():
    """k_da_bwd_unit_planes (round 4: the unit gradients of the DA backward with the forward's mapping -- head planes, a wave = one
    head of a patch of 64 queries) against k_da_cross_attn_bwd_unit on the same inputs: d/d offsets, d/d attention, d/d depth
    distribution equal up to fp32 re-association, the value gradient (whose fixed-point scale both kernels fold) bit for bit.
    8 x 8 patches of a grid that is not a multiple of the patch, runs of 64 consecutive queries, 2 / 4 levels with a 2-wide
    level, plain and chunk-major token rows."""
    import os
    cases = ((31, dict(B=2, Q=5 * 11, shapes=((16, 44), (8, 22))), 11),
             (32, dict(B=1, Q=9 * 8, shapes=((5, 7), (9, 6), (3, 4), (2, 2))), 8),
             (33, dict(B=1, Q=70, shapes=((6, 9),)), 0))
    for seed, kw, bev_w in cases:
        args, exp = _da_case(seed, E=80, M=8, P=8, DC=20, **kw)
        value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed))
        shapes_host = [tuple(int(x) for x in hw) for hw in ss.tolist()]
        Dh = value.shape[-1]
        vp = torch.zeros(value.shape[:-1] + (12,)); vp[..., :Dh] = value
        for hm, vin in ((0, vp), (5, _interleave(vp))):
            o_in = offsets.permute(0, 1, 3, 4, 2, 5).contiguous() if hm & 1 else offsets
            got = {}
            for planes in ('0', '1'):
                os.environ['FBBEV_DA_BWD_OWNED'] = '1'
                os.environ['FBBEV_DA_BWD_UNIT_PLANES'] = planes
                try:
                    got[planes] = E.da_cross_attn_bwd(vin, ss, ls, pred, ref_cam, mask, qdepth, o_in, attn, d0, dstep, g, head_minor=hm,
                                                      head_dim=Dh, lds_planes=True, level_hw=shapes_host, bev_w=bev_w)
                finally:
                    del os.environ['FBBEV_DA_BWD_OWNED'], os.environ['FBBEV_DA_BWD_UNIT_PLANES']
            # the training forward on the same planes (k_da_fwd_planes) against the unit kernel / the oracle composite
            code, slots, planes = E.da_cross_attn_fwd_planes(vin, ss, ls, pred, ref_cam, mask, qdepth, o_in, attn, d0, dstep, head_minor=hm,
                                                             head_dim=Dh, bev_w=bev_w)
            assert code == 0 and torch.equal(planes, value.permute(0, 2, 1, 3))
            assert not torch.isnan(slots).any()
            assert torch.allclose(slots, exp, atol=2e-5, rtol=1e-5), (seed, hm, (slots - exp).abs().max())
            if hm == 0:       # the planes route is planned: its workspace holds the (B*Ncam, M, S, Dh) planes behind the hit lists
                import ctypes
                flat = [int(x) for hw in shapes_host for x in hw]
                harr = (ctypes.c_int32 * len(flat))(*flat)
                Ncam_, B_, Q_, _ = mask.shape
                sizes = {}
                for planes in ('0', '1'):
                    os.environ['FBBEV_DA_BWD_OWNED'] = '1'
                    os.environ['FBBEV_DA_BWD_UNIT_PLANES'] = planes
                    sizes[planes] = E.lib().fbbev_da_cross_attn_bwd_ws_bytes_za(B_, Ncam_, vp.shape[1], 8, Dh, Q_, 12, len(shapes_host), 8, mask.shape[3], harr)
                    del os.environ['FBBEV_DA_BWD_OWNED'], os.environ['FBBEV_DA_BWD_UNIT_PLANES']
                plane_bytes = B_ * Ncam_ * 8 * vp.shape[1] * Dh * 4
                assert plane_bytes <= sizes['1'] - sizes['0'] < plane_bytes + 256, (sizes, plane_bytes)
            assert torch.equal(got['0'][0], got['1'][0]), (seed, hm)                      # value gradient: same scale, same bits
            for name, x, y in zip(('pred', 'offsets', 'attn'), got['1'][1:], got['0'][1:]):
                assert not torch.isnan(x).any()
                scale = y.abs().max().item()
                assert scale > 0 and (x - y).abs().max().item() <= 2e-6 * scale + 1e-7, (seed, hm, name, (x - y).abs().max().item(), scale)


@pytest.mark.parametrize('B,T1,C,N,dt', [(1, 3, 16, 64, torch.float32), (2, 2, 80, 100, torch.bfloat16), (1, 3, 80, 17, torch.float16),
                                         (1, 17, 80, 64, torch.bfloat16)])
def test_history_conv_bf16_mfma_emulated(B, T1, C, N, dt):
    """k_history_conv_bf16 on the emulated v_mfma_f32_16x16x32_bf16 against the same roundings restated in float64: weights and
    frames to bf16, fp32-exact products, relu(. + b1) rounded to bf16, second GEMM + b2, relu.  A bf16 rounding of the
    intermediate may land on the other neighbour when the fp32 accumulation order differs in the last bit: bounded, rare."""
    g = torch.Generator().manual_seed(N + T1)
    big = (torch.randn(B, T1 * C + 8, N, generator=g)).to(dt)
    feats = big[:, 8:]
    w1, w2 = torch.randn(C, C, generator=g) * 0.3, torch.randn(C, T1 * C, generator=g) * 0.2
    b1, b2 = torch.randn(B * T1, C, generator=g), torch.randn(C, generator=g)
    got = E.history_conv(feats, w1, b1, w2, b2, bf16=True)
    r = lambda t: t.bfloat16().double()  # noqa: E731
    x = r(feats.float()).reshape(B, T1, C, N)
    y = torch.relu(torch.einsum('oc,btcn->bton', r(w1), x) + b1.view(B, T1, C, 1).double())
    exp = torch.relu(torch.einsum('oc,bcn->bon', r(w2), r(y.float()).reshape(B, T1 * C, N)) + b2.view(1, C, 1).double())
    assert not torch.isnan(got).any()
    err = (got.double() - exp).abs()
    scale = exp.abs().max().item()
    assert err.max().item() <= 4e-3 * scale, (err.max().item(), scale)          # a few flipped bf16 neighbours of y at most
    assert err.mean().item() <= 2e-4 * scale
    # and the reduced precision itself stays where the docstring says: ~1e-2 of the fp32 kernel's volume
    ref = E.history_conv(feats, w1, b1, w2, b2)
    assert (got - ref).abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize('B,T1,C,Cout,N', [(1, 3, 16, 16, 64), (2, 2, 32, 16, 100), (1, 4, 16, 32, 17)])
def test_history_conv_mfma_emulated(B, T1, C, Cout, N):
    """k_history_conv on the emulated v_mfma_f32_16x16x4_f32: out = relu(b2 + sum_t W2_t relu(W1 x_t + b1_t))."""
    g = torch.Generator().manual_seed(N)
    big = torch.randn(B, T1 * C + 8, N, generator=g)
    feats = big[:, 8:]                                       # per-sample block contiguous, batch stride padded
    w1, w2 = torch.randn(C, C, generator=g) * 0.3, torch.randn(Cout, T1 * C, generator=g) * 0.2
    b1, b2 = torch.randn(B * T1, C, generator=g), torch.randn(Cout, generator=g)
    got = E.history_conv(feats, w1, b1, w2, b2)
    x = feats.reshape(B, T1, C, N).double()
    y = torch.relu(torch.einsum('oc,btcn->bton', w1.double(), x) + b1.view(B, T1, C, 1).double())
    exp = torch.relu(torch.einsum('oc,bcn->bon', w2.double(), y.reshape(B, T1 * C, N)) + b2.view(1, Cout, 1).double())
    assert not torch.isnan(got).any()
    assert torch.allclose(got.double(), exp, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('name,tv,flags', [('TINY', 64, 0), ('TINY', 128, 0x24), ('SMALL', 128, 0x24424), ('SMALL', 256, 0x4)])
def test_pool_zmean_and_add_epilogue_emulated(name, tv, flags):
    """fbbev_pool_zmean == mean over z of the pooled volume; fbbev_bev_pool_v2_dense_fwd_add == volume + addend[...,None]
    (fbocc.py:359,365-366): the volume is written once and never re-read."""
    B = 2 if name == 'TINY' else 1
    cfg, vt, coor, depth, feat = _case(name, B)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt))
    _, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    vol = O.bev_pool_v2(depth, feat, erd, erf, erb, (B, Z, Y, X, C), est, eln, use_fma=True)      # (B,C,Z,Y,X)
    code, mean = E.pool_zmean(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    assert code == 0 and not torch.isnan(mean).any()
    assert torch.allclose(mean, vol.mean(2), atol=1e-6, rtol=1e-5)
    assert torch.allclose(mean, vol.double().sum(2).float() / Z, atol=1e-6, rtol=1e-5)
    # round 5: the COLUMN form of the same entry (k_pool_zmean_col, FBBEV_ZMEAN_COL=1: all planes' metadata at once, a lane group
    # per pixel; opt-in, measured slower) -- the bits of the plane-after-plane walk; SMALL columns hold more points than the staging cap
    import os
    os.environ['FBBEV_ZMEAN_COL'] = '1'
    try:
        code, col = E.pool_zmean(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    finally:
        del os.environ['FBBEV_ZMEAN_COL']
    assert code == 0 and torch.equal(mean, col)
    # fbbev_pool_zmean_split: the planes of a tile dealt to 2 / 3 / Z workgroups + the ordered reduce (another association of the z sum)
    for zg in (2, 3, Z):
        code, m2 = E.pool_zmean(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags, z_groups=zg)
        assert code == 0 and not torch.isnan(m2).any(), zg
        assert torch.allclose(m2, vol.double().sum(2).float() / Z, atol=1e-6, rtol=1e-5), zg
        if zg == Z:
            assert torch.equal(m2, mean)                             # one plane per group: the single pass's bits
    # round 6: fbbev_pool_zmean_rows -- the mean written as the backward projection's query rows (B, Y*X, C) + a (Y*X, C) row bias
    # (bev_embedding): the bits of the planes form transposed, + one fp32 add
    emb = torch.randn(Y * X, C, generator=torch.Generator().manual_seed(5))
    code, rows = E.pool_zmean_rows(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
    assert code == 0 and torch.equal(rows, mean.flatten(2).transpose(1, 2))
    code, rows = E.pool_zmean_rows(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags, row_bias=emb)
    assert code == 0 and torch.equal(rows, mean.flatten(2).transpose(1, 2) + emb[None])
    addend = torch.randn(B, C, Y, X, generator=torch.Generator().manual_seed(3))
    code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags, addend=addend)
    assert code == 0
    assert torch.equal(out, vol + addend[:, :, None])                # one fp32 add per element: identical bits
    if (Y * X) % 8 == 0:
        code, o16 = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags | 0x800000,
                                 addend=addend)
        assert code == 0 and torch.equal(o16.view(torch.int16), (vol + addend[:, :, None]).to(torch.bfloat16).view(torch.int16))


def test_rank_build_with_depth_threshold_emulated():
    """BEVDet-era filter kept &= depth > 0.01 (mmdet3d/models/necks/view_transformer.py:552-557): data-dependent P."""
    cfg, vt, coor, depth, feat = _case('TINY', 2)
    g = torch.Generator().manual_seed(9)
    depth = depth.clone()
    depth[torch.rand(depth.shape, generator=g) < 0.4] = 0.005                  # 40 % of the points fall under the threshold
    depth.view(-1)[0] = 0.01                                                   # exactly the threshold: dropped (strict >)
    rb, rd, rf, st, ln, ir, counts = E.rank_build(coor, *_grid3(vt), depth=depth.contiguous(), depth_threshold=0.01)
    # oracle: the reference prepare on the coordinates with the under-threshold points pushed out of the grid
    c2 = coor.clone()
    c2.view(-1, 3)[~(depth.reshape(-1) > 0.01)] = 1.0e6
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(c2)
    P, I = counts.tolist()
    assert (P, I) == (erb.numel(), est.numel()) and P < E.rank_build(coor, *_grid3(vt))[6][0]
    assert torch.equal(rb[:P], erb) and torch.equal(rd[:P], erd) and torch.equal(rf[:P], erf)
    assert torch.equal(st[:I], est) and torch.equal(ln[:I], eln)


@pytest.mark.parametrize('B,Q,M,Dh,shapes,P', [(2, 31, 4, 10, [[6, 5]], 4), (1, 17, 2, 8, [[5, 4], [3, 2]], 3), (1, 9, 2, 4, [[4, 4]], 2)])
def test_msda_fwd_fused_equals_unfused_emulated(B, Q, M, Dh, shapes, P):
    """fbbev_msda_fwd_fused (loc = ref + offset / size inside the kernel, unit-per-lane, head-padded rows) produces the
    bits of fbbev_msda_fwd on the location tensor torch builds the way mmcv does."""
    g = torch.Generator().manual_seed(Q)
    ss = torch.tensor(shapes)
    ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    S_, L = int((ss[:, 0] * ss[:, 1]).sum()), len(shapes)
    value = torch.randn(B, S_, M, Dh, generator=g)
    ref = torch.rand(B, Q, L, 2, generator=g)
    so = torch.randn(B, Q, M, L, P, 2, generator=g) * 2.0
    w = torch.rand(B, Q, M, L * P, generator=g).softmax(-1).view(B, Q, M, L, P).contiguous()
    norm = torch.stack([ss[..., 1], ss[..., 0]], -1)
    loc = (ref[:, :, None, :, None, :] + so / norm[None, None, None, :, None, :]).contiguous()      # mmcv's two passes
    base = E.msda_fwd(value, ss, ls, loc, w)
    assert torch.equal(E.msda_fwd_fused(value, ss, ls, ref, so, w), base)
    assert torch.equal(E.msda_fwd_fused(value, ss, ls, ref, so.permute(0, 1, 3, 4, 2, 5).contiguous(), w,
                                        offsets_head_minor=True), base)
    HS = (Dh + 3) // 4 * 4 + (4 if Dh % 4 == 0 else 0)
    vp = torch.full((B, S_, M, HS), -3.0e4)
    vp[..., :Dh] = value
    assert torch.equal(E.msda_fwd_fused(vp, ss, ls, ref, so, w, head_dim=Dh), base)
    assert torch.equal(E.msda_fwd_fused(_interleave(vp), ss, ls, ref, so, w, head_dim=Dh, value_interleaved=True), base)


# ---------------------------------------------------------------- one-launch-per-pass sort: chunk shapes + look-back
def _lift_vs_oracle(name, B, cache=None, cam=None):
    cfg = S.CONFIGS[name]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = cam if cam is not None else S.camera_rig(cfg, B, seed=0, bda_aug=True)
    xs = vt.frustum[0, 0, :, 0].contiguous(); ys = vt.frustum[0, :, 0, 1].contiguous(); ds = vt.frustum[:, 0, 0, 2].contiguous()
    got = E.lift_rank_build(xs, ys, ds, cam, *_grid3(vt), frustum=vt.frustum.contiguous(), cache=cache)
    coor = E.lidar_coor(xs, ys, ds, cam)                     # the contract pins bit-exactness at coor
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    return got, (erb, erd, erf, est, eln)


def _assert_index_equal(got, exp):
    rb, rd, rf, st, ln, ir, counts = got
    erb, erd, erf, est, eln = exp
    P, I = counts.tolist()
    assert (P, I) == (erb.numel(), est.numel())
    assert torch.equal(rb[:P], erb) and torch.equal(rd[:P], erd) and torch.equal(rf[:P], erf)
    assert torch.equal(st[:I], est) and torch.equal(ln[:I], eln) and torch.equal(ir[:I], erb[est.long()])


@pytest.mark.parametrize('shape', ['0,0,0', '1,0,1', '2,1,2', '3,2,1', '4,3,2', '0,4,0', '5,6,1', '6,5,2'])
def test_rank_build_every_chunk_variant_emulated(shape, monkeypatch):
    """Every chunk shape of the sort / interval kernels (thin 256-thread chunks up to 16 waves x 16 rounds) on the SAME
    input: the index tensors do not depend on the shape (FBBEV_RANK_SHAPE = the launcher's tuning knob) and equal the
    oracle.  SMALL B=2 (61 k points): several chunks per pass for the thin shapes, partial chunks for the fat ones."""
    monkeypatch.setenv('FBBEV_RANK_SHAPE', shape)
    got, exp = _lift_vs_oracle('SMALL', 2)
    _assert_index_equal(got, exp)


@pytest.mark.parametrize('name,B', [('TINY', 2), ('SMALL', 2), ('SMALL', 3), ('REF', 1)])
def test_segmented_sort_equals_global_sort_and_oracle_emulated(name, B, monkeypatch):
    """Round 3: the per-sample (segmented) sort -- sample-aligned chunks, digits of (key - b V), two passes of <= 10 bits, 6
    launches -- against the global 8-bit LSD sort and the oracle: the same index tensors, bit for bit, through BOTH key
    sources (camera geometry and a materialised coor), partial last chunks and one- and two-pass digit plans included."""
    monkeypatch.setenv('FBBEV_RANK_SEG', '0')
    glob_, exp = _lift_vs_oracle(name, B)
    _assert_index_equal(glob_, exp)
    monkeypatch.setenv('FBBEV_RANK_SEG', '2')
    seg, _ = _lift_vs_oracle(name, B)
    _assert_index_equal(seg, exp)
    P, I = seg[6].tolist()
    for a, b_ in zip(seg[:3], glob_[:3]):
        assert torch.equal(a[:P], b_[:P])
    for a, b_ in zip(seg[3:6], glob_[3:6]):
        assert torch.equal(a[:I], b_[:I])
    # the two-step contract path (keys from coor)
    cfg, vt, coor, depth, feat = _case(name, B)
    got = E.rank_build(coor, *_grid3(vt))
    erb, erd, erf, est, eln = vt.voxel_pooling_prepare_v2(coor)
    _assert_index_equal(got, (erb, erd, erf, est, eln))


def test_segmented_sort_with_depth_filter_and_empty_samples_emulated(monkeypatch):
    """Data-dependent P (BEVDet-era `depth > 0.01` filter) and a sample without a single kept point: segment ranges of
    length 0, chunks that return at once, count-matrix rows of zeros."""
    monkeypatch.setenv('FBBEV_RANK_SEG', '2')
    cfg, vt, coor, depth, feat = _case('SMALL', 3)
    coor = coor.clone()
    coor[1] = 1.0e6                                           # sample 1: every point outside the grid
    d = depth.clone()
    d[d < d.median()] = 0.0
    got = E.rank_build(coor, *_grid3(vt), depth=d, depth_threshold=0.01)
    monkeypatch.setenv('FBBEV_RANK_SEG', '0')
    ref = E.rank_build(coor, *_grid3(vt), depth=d, depth_threshold=0.01)
    P, I = got[6].tolist()
    assert (P, I) == tuple(ref[6].tolist()) and P > 0
    for a, b_ in zip(got[:3], ref[:3]):
        assert torch.equal(a[:P], b_[:P])
    for a, b_ in zip(got[3:6], ref[3:6]):
        assert torch.equal(a[:I], b_[:I])
    assert not ((got[0][:P] >= 20000) & (got[0][:P] < 40000)).any()      # nothing of sample 1 (SMALL: 50x50x8 voxels per sample)


def test_rank_build_fat_chunks_at_bench_scale_emulated():
    """n = 0.75 M points (BL2, 3 samples): the launcher itself picks 16-wave chunks for pass 0 and the 1024-thread interval
    kernels -- many full chunks per pass, bit-exact against the oracle on the emulator."""
    got, exp = _lift_vs_oracle('BL2', 3)
    _assert_index_equal(got, exp)


def test_camera_keyed_cache_skips_and_rebuilds_emulated():
    """same rig -> the build is skipped on the device (state[1] counts builds), a changed bda -> rebuild,
    bit-exact with a fresh build; the skip leaves the previous index set untouched."""
    cfg = S.CONFIGS['TINY']
    cache = {}
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
    got, exp = _lift_vs_oracle('TINY', 2, cache=cache, cam=cam)
    _assert_index_equal(got, exp)
    assert cache['state'].tolist() == [0, 1]
    snap = [t.clone() for t in got]
    got2, _ = _lift_vs_oracle('TINY', 2, cache=cache, cam=[t.clone() for t in cam])
    assert cache['state'].tolist() == [1, 1]                 # skipped: nothing ran
    assert all(torch.equal(a, b) for a, b in zip(got2, snap))
    cam2 = [t.clone() for t in cam]
    cam2[5][1] = cam2[5][1] @ torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]])     # rotate sample 1's bda
    got3, exp3 = _lift_vs_oracle('TINY', 2, cache=cache, cam=cam2)
    assert cache['state'].tolist() == [0, 2]
    _assert_index_equal(got3, exp3)
    assert not torch.equal(got3[0][:got3[6][0]], snap[0][:got3[6][0]]) or got3[6].tolist() != snap[6].tolist()
    cam3 = [t.clone() for t in cam2]
    cam3[4][0, 3, 1] += 1.0                                   # one post_trans element of one camera
    got4, exp4 = _lift_vs_oracle('TINY', 2, cache=cache, cam=cam3)
    assert cache['state'].tolist() == [0, 3]
    _assert_index_equal(got4, exp4)


def test_cached_tile_table_is_gated_per_table_emulated():
    """a cache hit may keep a tile table only if THAT table was built for the current index set.
    Two tables (tile 64 / 128) of one cached set: the table that missed the last rebuild, and a freshly allocated one,
    are rebuilt on the next hit although cache_state[0] == 1; a table that is current is left untouched."""
    from ctypes import c_void_p
    cfg = S.CONFIGS['TINY']
    cache = {}
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
    got, _ = _lift_vs_oracle('TINY', 2, cache=cache, cam=cam)
    rb, rd, rf, st, ln, ir, counts = got
    X, Y, Z = cfg.grid_xyz
    lib = E.lib()
    nbytes = lib.fbbev_pool_dense_workspace_bytes(2, Z, Y, X)

    def table():
        return torch.full((nbytes,), 0x5A, dtype=torch.uint8), torch.full((2,), -1, dtype=torch.int32)

    def build(tbl, gate, tv, state):
        E.ok(lib.fbbev_pool_tile_index_cached(E.p(ir), E.p(st), E.p(counts), ir.numel(), 2, Z, Y, X, tv, 0, E.p(tbl),
                                              tbl.numel(), E.p(state), E.p(gate), None))

    def used(tv):                                                # bytes the table kernel writes: (tiles + 1) x 2 ints
        return (2 * Z * ((Y * X + tv - 1) // tv) + 1) * 8

    def fresh(tv):
        t = torch.full((nbytes,), 0x5A, dtype=torch.uint8)
        E.ok(lib.fbbev_pool_tile_index(E.p(ir), E.p(st), E.p(counts), ir.numel(), 2, Z, Y, X, tv, 0, E.p(t), t.numel(), None))
        return t[:used(tv)]

    state = cache['state']
    assert state.tolist() == [0, 1]
    t64, g64 = table()
    build(t64, g64, 64, state)                                   # build 1 ran: table built, gate remembers build 1
    assert g64.tolist() == [1, 0] and torch.equal(t64[:used(64)], fresh(64))
    _lift_vs_oracle('TINY', 2, cache=cache, cam=[t.clone() for t in cam])
    assert state.tolist() == [1, 1]                              # hit
    t64.fill_(0x11)
    build(t64, g64, 64, state)
    assert g64.tolist() == [1, 1] and int(t64[0]) == 0x11        # current table: kept (not rewritten)
    t128, g128 = table()                                         # allocated AFTER the build: must not be trusted on a hit
    build(t128, g128, 128, state)
    assert g128.tolist() == [1, 0] and torch.equal(t128[:used(128)], fresh(128))
    # rebuild with another rig through table 128 only; the next hit through table 64 must rebuild table 64
    cam2 = [t.clone() for t in cam]
    cam2[5][1] = cam2[5][1] @ torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]])
    got2, _ = _lift_vs_oracle('TINY', 2, cache=cache, cam=cam2)
    rb, rd, rf, st, ln, ir, counts = got2
    assert state.tolist() == [0, 2]
    build(t128, g128, 128, state)
    assert g128.tolist() == [2, 0] and torch.equal(t128[:used(128)], fresh(128))
    _lift_vs_oracle('TINY', 2, cache=cache, cam=[t.clone() for t in cam2])
    assert state.tolist() == [1, 2]
    stale = t64.clone()
    build(t64, g64, 64, state)                                   # hit, but table 64 belongs to build 1
    assert g64.tolist() == [2, 0] and torch.equal(t64[:used(64)], fresh(64)) and not torch.equal(t64, stale)


# ---------------------------------------------------------------- 16-bit storage of the history ring (BASELINE configs[4])
@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_history_warp_and_conv_16bit_storage_emulated(dt):
    """fbbev_history_warp_e / fbbev_history_conv_e: the stored elements are widened exactly, the arithmetic is the fp32
    kernel's, the warp result is rounded once (nearest-even) at the store -- i.e. bit-identical to the fp32 kernel run on
    the widened input followed by torch's own fp32 -> 16-bit conversion."""
    g = torch.Generator().manual_seed(9)
    B, CH, Z, Y, X = 2, 10, 3, 8, 12
    hist = (torch.randn(B, CH, Z, Y, X, generator=g) * 3).to(dt)
    hist[0, 0, 0, 0, :4] = torch.tensor([0.0, -0.0, 6.0e-5, 65000.0]).to(dt)     # zero, signed zero, subnormal half, near max
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    got = E.history_warp(hist, flow)
    assert got.dtype == dt
    exp32 = E.history_warp(hist.float(), flow)                                   # same kernel arithmetic on the widened taps
    assert torch.equal(got.view(torch.int16), exp32.to(dt).view(torch.int16))
    ident = E.history_warp(hist, torch.eye(4)[None].repeat(B, 1, 1).contiguous())
    # identity flow: the stored bits survive (sample 1; sample 0 holds the 65000 next to small values, where the 1e-7
    # residual tap weight of the reference's normalise / un-normalise round trip is visible in half precision)
    assert torch.equal(ident[1].view(torch.int16), hist[1].view(torch.int16))
    # convolution reading a 16-bit frame buffer == the fp32 kernel on the widened buffer, bit for bit
    T1, C, Cout, N = 3, 16, 16, 70
    feats = (torch.randn(B, T1 * C, N, generator=g)).to(dt)
    w1, w2 = torch.randn(C, C, generator=g) * 0.3, torch.randn(Cout, T1 * C, generator=g) * 0.2
    b1, b2 = torch.randn(B * T1, C, generator=g), torch.randn(Cout, generator=g)
    assert torch.equal(E.history_conv(feats, w1, b1, w2, b2), E.history_conv(feats.float(), w1, b1, w2, b2))
    T1, C, Cout = 2, 32, 16                                                      # the generic (non register-resident) kernel
    feats = (torch.randn(B, T1 * C, N, generator=g)).to(dt)
    w1, w2 = torch.randn(C, C, generator=g) * 0.3, torch.randn(Cout, T1 * C, generator=g) * 0.2
    b1 = torch.randn(B * T1, C, generator=g)
    assert torch.equal(E.history_conv(feats, w1, b1, w2, b2), E.history_conv(feats.float(), w1, b1, w2, b2))


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('producers', ['8', '5'])
def test_history_fused_warp_and_conv_equals_the_two_kernels_emulated(dt, producers, monkeypatch):
    """fbbev_history_fused_vm (one launch: producer waves warp the previous ring into the next one and into an LDS operand
    tile, consumer waves run both bf16-MFMA convolutions from it) against fbbev_history_warp_vm + fbbev_history_conv_bf16 on
    the same rings: the new ring's slots 1..T and the fused volume are the SAME BITS -- same taps, weights, roundings, operands
    and accumulation order; only where the operands come from differs.  Grid rows that are not a multiple of the 64-voxel
    tile, translation / rotation / out-of-grid / NaN flows, padded batch strides, both producer counts."""
    monkeypatch.setenv('FBBEV_HISTORY_FUSED_PRODUCERS', producers)
    g = torch.Generator().manual_seed(13)
    B, T, C, Z, Y, X = 4, 3, 80, 2, 3, 70                                     # X = 70: one full tile + a 6-voxel tile per row
    N = Z * Y * X
    hist = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)              # spare slot: padded batch stride
    hist[:, :T] = (torch.randn(B, T, N, C, generator=g) * 2).to(dt)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    flow[2, :3, 3] = torch.tensor([500.0, 0.0, 0.0])                         # leaves the grid: zero padding
    flow[3, 0, 0] = float('nan')
    curr = torch.randn(B, C, N, generator=g)
    w1, w2 = torch.randn(C, C, generator=g) * 0.2, torch.randn(C, (T + 1) * C, generator=g) * 0.1
    b1, b2 = torch.randn(B * (T + 1), C, generator=g), torch.randn(C, generator=g)
    # the two-kernel path
    ref = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_frame_vm(curr, dt, out=ref[:, 0])
    E.history_warp_vm(hist[:, :T], flow, (Z, Y, X), out=ref[:, 1:])
    exp = E.history_conv(ref, w1, b1, w2, b2, bf16=True, voxel_major=True)
    # the fused kernel
    nxt = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_frame_vm(curr, dt, out=nxt[:, 0])
    code, got = E.history_fused_vm(hist[:, :T], flow, nxt, (Z, Y, X), w1, b1, w2, b2)
    assert code == 0
    assert torch.equal(nxt.view(torch.int16), ref.view(torch.int16))         # the ring: identical element bits (NaN flows included)
    fin = torch.isfinite(exp)
    assert torch.equal(torch.isfinite(got), fin) and torch.equal(got[fin], exp[fin])
    assert fin.all()                                                         # a NaN flow samples nothing (zero taps), it does not poison


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_history_fused_x3_equals_the_two_kernels_emulated(dt):
    """fbbev_history_fused_x3_vm (one launch: every MFMA wave blends the taps of its own operands, stores them to the next ring
    and runs both split-operand convolutions on them) against fbbev_history_warp_vm + fbbev_history_conv_bf16x3 on the same
    rings: slots 1..T of the new ring and the fused volume are the SAME BITS.  Bricks that overhang the grid in x (70 = 4 x 16 + 6)
    and y (11 = 8 + 3), translation / rotation / out-of-grid / NaN flows, a padded batch stride, -0.0 in the current frame."""
    g = torch.Generator().manual_seed(23)
    B, T, C, Z, Y, X = 4, 3, 80, 2, 11, 70
    N = Z * Y * X
    hist = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)              # spare slot: padded batch stride
    hist[:, :T] = (torch.randn(B, T, N, C, generator=g) * 2).to(dt)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    flow[2, :3, 3] = torch.tensor([500.0, 0.0, 0.0])                         # leaves the grid: zero padding
    flow[3, 0, 0] = float('nan')
    curr = torch.randn(B, C, N, generator=g)
    curr[0, :, :5] = -0.0
    w1, w2 = torch.randn(C, C, generator=g) * 0.2, torch.randn(C, (T + 1) * C, generator=g) * 0.1
    b1, b2 = torch.randn(B * (T + 1), C, generator=g), torch.randn(C, generator=g)
    ref = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_frame_vm(curr, dt, out=ref[:, 0])
    E.history_warp_vm(hist[:, :T], flow, (Z, Y, X), out=ref[:, 1:])
    exp = E.history_conv(ref, w1, b1, w2, b2, voxel_major=True, x3=True)
    nxt = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_frame_vm(curr, dt, out=nxt[:, 0])
    code, got = E.history_fused_x3_vm(hist[:, :T], flow, nxt, (Z, Y, X), w1, b1, w2, b2)
    assert code == 0
    assert torch.equal(nxt.view(torch.int16), ref.view(torch.int16))         # the ring: identical element bits (NaN flows included)
    assert torch.isfinite(exp).all() and torch.equal(got, exp)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_history_step_in_row_bands_equals_the_two_calls_emulated(dt):
    """fbbev_history_step_x3_vm (warp and split-operand convolutions launched band of rows by band of rows: the chunks of the
    two-stream pipeline) against fbbev_history_warp_vm + fbbev_history_conv_bf16x3 over the whole volume: the ring and the fused
    volume are the SAME BITS for 1 chunk (back to back), 2, and more chunks than bands; bands whose row segments are not a
    multiple of the 256-voxel tile, a padded batch stride, translation / rotation / out-of-grid flows."""
    g = torch.Generator().manual_seed(17)
    B, T, C, Z, Y, X = 3, 2, 80, 2, 5, 37                                     # YB = 64 rows per band -> clamped to Y; see FBBEV_HISTORY_VM_YB below
    N = Z * Y * X
    hist = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)              # spare slot: padded batch stride
    hist[:, :T] = (torch.randn(B, T, N, C, generator=g) * 2).to(dt)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    flow[2, :3, 3] = torch.tensor([500.0, 0.0, 0.0])
    curr = torch.randn(B, C, N, generator=g)
    w1, w2 = torch.randn(C, C, generator=g) * 0.2, torch.randn(C, (T + 1) * C, generator=g) * 0.1
    b1, b2 = torch.randn(B * (T + 1), C, generator=g), torch.randn(C, generator=g)
    ref = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_frame_vm(curr, dt, out=ref[:, 0])
    E.history_warp_vm(hist[:, :T], flow, (Z, Y, X), out=ref[:, 1:])
    exp = E.history_conv(ref, w1, b1, w2, b2, voxel_major=True, x3=True)
    assert torch.isfinite(exp).all()
    import os
    os.environ['FBBEV_HISTORY_VM_YB'] = '2'                                  # 3 bands of rows: 2 + 2 + 1
    try:
        for chunks in (1, 2, 0, 64):
            nxt = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
            E.history_frame_vm(curr, dt, out=nxt[:, 0])
            got = E.history_step_x3_vm(hist[:, :T], flow, nxt, (Z, Y, X), w1, b1, w2, b2, chunks=chunks)
            assert torch.equal(nxt.view(torch.int16), ref.view(torch.int16)), chunks
            assert torch.equal(got, exp), chunks
    finally:
        del os.environ['FBBEV_HISTORY_VM_YB']


@pytest.mark.parametrize('dt', [torch.float32, torch.float16, torch.bfloat16])
def test_history_voxel_major_ring_equals_planar_kernels_emulated(dt):
    """The voxel-major ring ([T][N][C] frames): fbbev_history_frame_vm is the rounded transpose of a frame,
    fbbev_history_warp_vm produces the planar kernel's elements bit for bit (translation, rotation, out-of-grid, NaN flow;
    padded batch stride; an odd frame count for the two-frames-per-thread loop), and fbbev_history_conv_bf16 reads either
    layout to the same bits."""
    g = torch.Generator().manual_seed(11)
    B, T, C, Z, Y, X = 4, 3, 16, 3, 7, 9
    N = Z * Y * X
    planar = (torch.randn(B, T * C, Z, Y, X, generator=g) * 2).to(dt)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    flow[2, :3, 3] = torch.tensor([50.0, 0.0, 0.0])                          # leaves the grid: zero padding
    flow[3, 0, 0] = float('nan')
    exp = E.history_warp(planar, flow)                                       # (B, T*C, Z, Y, X)
    big = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)               # ring with a spare slot: padded batch stride
    big[:, :T] = planar.view(B, T, C, N).transpose(2, 3)
    out = torch.full((B, T + 1, N, C), float('nan'), dtype=dt)
    E.history_warp_vm(big[:, :T], flow, (Z, Y, X), out=out[:, 1:])
    got = out[:, 1:].transpose(2, 3).reshape(B, T * C, Z, Y, X)
    bits = torch.int32 if dt == torch.float32 else torch.int16
    assert torch.equal(got.contiguous().view(bits), exp.view(bits))
    assert torch.isnan(out[:, 0].float()).all()                              # slot 0 untouched
    # slot 0: the current frame, transposed and rounded once
    curr = torch.randn(B, C, N, generator=g) * 3
    curr[0, 0, :4] = torch.tensor([0.0, -0.0, 6.0e-5, 65000.0])
    E.history_frame_vm(curr, dt, out=out[:, 0])
    assert torch.equal(out[:, 0].contiguous().view(bits), curr.transpose(1, 2).to(dt).contiguous().view(bits))
    N2 = 70                                                                   # a partial 64-voxel tile
    c2 = torch.randn(2, 80, N2, generator=g)
    assert torch.equal(E.history_frame_vm(c2, dt).view(bits), c2.transpose(1, 2).to(dt).contiguous().view(bits))
    # a (Y, X, Z) volume -> (Z, Y, X)-ordered rows
    vol = torch.randn(2, 16, 5, 7, 3, generator=g)                            # B, C, Y, X, Z
    got = E.history_frame_vm(vol.view(2, 16, -1), dt, inner=3)
    assert torch.equal(got.view(bits), vol.permute(0, 4, 2, 3, 1).reshape(2, -1, 16).to(dt).contiguous().view(bits))
    # a single frame (the two-frames-per-thread batch clamps to it), C = 80
    one = (torch.randn(1, 1, 80, N, generator=g)).to(dt)
    f1 = torch.eye(4)[None].clone(); f1[0, :3, 3] = torch.tensor([0.5, 0.25, -0.5])
    got1 = E.history_warp_vm(one.transpose(2, 3).contiguous(), f1, (Z, Y, X))
    exp1 = E.history_warp(one.reshape(1, 80, Z, Y, X), f1)
    assert torch.equal(got1.transpose(2, 3).reshape(1, 80, Z, Y, X).contiguous().view(bits), exp1.view(bits))
    # the convolutions read rows: same bits as from planes (T1 = 1: the 3-slot prefetch ring with a single frame)
    for Cc, T1, n in ((16, 3, 70), (80, 2, 33), (16, 1, 20), (80, 5, 16)):
        feats = torch.randn(2, T1, Cc, n, generator=g).to(dt)
        w1, w2 = torch.randn(Cc, Cc, generator=g) * 0.3, torch.randn(Cc, T1 * Cc, generator=g) * 0.2
        b1, b2 = torch.randn(2 * T1, Cc, generator=g), torch.randn(Cc, generator=g)
        a = E.history_conv(feats.reshape(2, T1 * Cc, n), w1, b1, w2, b2, bf16=True)
        b = E.history_conv(feats.transpose(2, 3).contiguous(), w1, b1, w2, b2, bf16=True, voxel_major=True)
        assert not torch.isnan(a).any() and torch.equal(a, b)
        # fp32 MFMA on rows: the same products, K summed in the voxel-major slot order -> equal to fp32 rounding
        a32 = E.history_conv(feats.reshape(2, T1 * Cc, n), w1, b1, w2, b2)
        b32 = E.history_conv(feats.transpose(2, 3).contiguous(), w1, b1, w2, b2, voxel_major=True)
        x = feats.double()
        y = torch.relu(torch.einsum('oc,btcn->bton', w1.double(), x) + b1.view(2, T1, Cc, 1).double())
        exp = torch.relu(torch.einsum('oc,bcn->bon', w2.double(), y.reshape(2, T1 * Cc, n)) + b2.view(1, Cc, 1).double())
        assert not torch.isnan(b32).any()
        assert torch.allclose(b32.double(), exp, atol=2e-5, rtol=1e-5) and torch.allclose(a32, b32, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_history_convs_bf16x3_are_fp32_grade_emulated(dt):
    """fbbev_history_conv_bf16x3 (operands split into two bf16 terms, three MFMAs per product) on a 16-bit voxel-major ring against
    the float64 convolutions of the SAME stored frames: < 2e-5 of the output peak (measured 5e-6) -- more than 50x better than the
    plain bf16 route on the same input (2.6e-3), the fp32-MFMA kernel sits at 2e-7; partial 128-voxel tiles, T1 = 1 (prefetch ring with one frame)."""
    g = torch.Generator().manual_seed(31)
    for Cc, T1, n in ((16, 3, 70), (80, 2, 33), (16, 1, 20), (80, 5, 150)):
        feats = (torch.randn(2, T1, n, Cc, generator=g) * 2).to(dt)
        w1, w2 = torch.randn(Cc, Cc, generator=g) * 0.3, torch.randn(Cc, T1 * Cc, generator=g) * 0.2
        b1, b2 = torch.randn(2 * T1, Cc, generator=g), torch.randn(Cc, generator=g)
        got = E.history_conv(feats, w1, b1, w2, b2, voxel_major=True, x3=True)
        x = feats.double().transpose(2, 3)                                            # (B, T1, C, n)
        y = torch.relu(torch.einsum('oc,btcn->bton', w1.double(), x) + b1.view(2, T1, Cc, 1).double())
        exp = torch.relu(torch.einsum('oc,bcn->bon', w2.double(), y.reshape(2, T1 * Cc, n)) + b2.view(1, Cc, 1).double())
        peak = exp.abs().max()
        assert not torch.isnan(got).any()
        err3 = (got.double() - exp).abs().max() / peak
        err1 = (E.history_conv(feats, w1, b1, w2, b2, bf16=True, voxel_major=True).double() - exp).abs().max() / peak
        err32 = (E.history_conv(feats, w1, b1, w2, b2, voxel_major=True).double() - exp).abs().max() / peak
        assert err3 < 2e-5 and err3 < err1 / 50 and err32 < err3, (Cc, T1, n, float(err3), float(err1), float(err32))


@pytest.mark.parametrize('dt', [torch.float32, torch.float16])
def test_history_warp_lds_staged_equals_gather_kernel_emulated(dt, monkeypatch):
    """k_history_warp_lds (a brick's source box staged in LDS) == k_history_warp (8 global gathers per output), bit for bit:
    translation, yaw rotation + translation, a flip (mirrored box), a flow that leaves the grid (empty box, zero padding),
    a large rotation (box does not fit: per-tap global reads), NaN flow; partial bricks at every border."""
    g = torch.Generator().manual_seed(7)
    B, CH, Z, Y, X = 6, 5, 9, 37, 70                                        # 9 planes: two z-bricks; 70: a partial x-brick
    hist = torch.randn(B, CH, Z, Y, X, generator=g).to(dt)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([2.5, -1.25, 0.5])
    c, s = np.cos(0.03), np.sin(0.03)
    flow[1, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    flow[1, :3, 3] = torch.tensor([1.7, -0.6, -0.2])
    flow[2, 0, 0] = -1.0; flow[2, 0, 3] = X - 1.0                           # flip x (bda flip)
    flow[3, :3, 3] = torch.tensor([500.0, 0.0, 0.0])                        # everything outside
    c, s = np.cos(0.9), np.sin(0.9)
    flow[4, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    flow[4, :3, 3] = torch.tensor([20.0, -10.0, 0.0])                       # 52 degrees: the box of a brick does not fit
    flow[5, 1, 1] = float('nan')
    monkeypatch.setenv('FBBEV_HISTORY_WARP', 'direct')
    ref = E.history_warp(hist, flow)
    monkeypatch.setenv('FBBEV_HISTORY_WARP', 'lds')
    got = E.history_warp(hist, flow)
    assert not torch.isnan(got.float()).any()
    assert torch.equal(got.view(torch.int16 if dt == torch.float16 else torch.int32), ref.view(torch.int16 if dt == torch.float16 else torch.int32))
    assert (got[3] == 0).all() and (got[5] == 0).all() and got[0].abs().sum() > 0 and got[4].abs().sum() > 0
    big = torch.full((B, CH + 3, Z, Y, X), float('nan')).to(dt)             # strided output (channel slice of a ring)
    E.history_warp(hist, flow, big[:, 3:])
    assert torch.equal(big[:, 3:].contiguous().view(torch.int16 if dt == torch.float16 else torch.int32),
                       ref.view(torch.int16 if dt == torch.float16 else torch.int32)) and torch.isnan(big[:, :3].float()).all()


def test_one_kernel_da_cross_attention_emulated():
    """fbbev_da_cross_attn_fused -> k_da_cross_attn_fused (round 4): query rows -> slots in one kernel -- the sampling_offsets /
    attention_weights projections on the split-operand bf16 MFMA inside the workgroup, softmax in LDS, head-plane camera tokens,
    a wave = one head of an 8 x 8 patch of BEV queries -- against the oracle's composite (spatial_cross_attention_depth.py:136-223,
    513-595) and against the unit kernel fed with fp32 projections.  Grids that are / are not multiples of the patch, 1 / 2 / 4
    levels, a level only two tokens wide, with and without the positional addend; fbbev_rows_linear_x3_planes writes the same
    planes as the row-major projection re-laid out; unsupported shapes are refused."""
    cases = ((21, dict(B=1, Q=8 * 8, shapes=((6, 9),)), 8),                                  # one full patch, one level
             (22, dict(B=2, Q=5 * 11, shapes=((16, 44), (8, 22))), 11),                       # partial patches in x and y
             (23, dict(B=1, Q=9 * 8, shapes=((5, 7), (9, 6), (3, 4), (2, 2))), 8),            # 4 levels (LP = 32), a 2-wide level
             (25, dict(B=1, Q=8 * 8, shapes=((5, 7), (4, 6), (3, 4))), 8),                    # 3 levels: a head's logits straddle MFMA tiles
             (26, dict(B=1, N=3, Q=8 * 8, shapes=((6, 9), (3, 4))), 8),                       # fewer records than threads: the hit-flag batches' clamps
             (27, dict(B=2, N=9, Q=5 * 11, shapes=((6, 9), (3, 4))), 11))                     # nine cameras: hit masks beyond a byte, 2.25 flag batches
    import os
    for seed, kw, bev_w in cases:
        args, exp, ex = _da_case(seed, E=80, M=8, P=8, DC=20, extras=True, **kw)
        value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        BN, S_, M, Dh = value.shape
        Pm = ex['Pm']
        pre = 'a.deformable_attention.'
        planes = E.rows_to_head_planes(value.reshape(BN * S_, M * Dh).contiguous(), S_, M, Dh)
        assert torch.equal(planes, value.permute(0, 2, 1, 3))
        # value_proj straight into planes (split-operand arithmetic: ~1e-5 relative)
        key = ex['key']                                                               # (N, S, B, E)
        x = key.permute(2, 0, 1, 3).reshape(BN * S_, M * Dh).contiguous()
        code, vp = E.rows_linear_x3_planes(x, Pm[pre + 'value_proj.weight'].contiguous(), Pm[pre + 'value_proj.bias'].contiguous(), S_, M, Dh)
        assert code == 0 and not torch.isnan(vp).any()
        assert torch.allclose(vp, planes, atol=2e-5 * planes.abs().max().item(), rtol=0)
        unit = E.da_cross_attn_fwd(*args)                                             # fp32 projections, unit kernel
        for with_pos in (True, False):
            q = (ex['query'] if with_pos else ex['query'] + ex['qpos']).contiguous()
            B, Q = q.shape[:2]
            add = None
            if with_pos:      # per-row addend (period B*Q): the general form; a (Q, E) table repeats with period Q
                add = ex['qpos'].reshape(B * Q, -1).contiguous()
            both = []
            for hw in ('8', '4'):      # heads per workgroup: one 512-thread workgroup per patch / two 256-thread ones (the default)
                os.environ['FBBEV_DA_FUSED_HW'] = hw
                try:
                    code, slots = E.da_cross_attn_fused(planes, ss, ls, pred, ref_cam, mask, qdepth, q, add,
                                                        Pm[pre + 'sampling_offsets.weight'].contiguous(), Pm[pre + 'sampling_offsets.bias'].contiguous(),
                                                        Pm[pre + 'attention_weights.weight'].contiguous(), Pm[pre + 'attention_weights.bias'].contiguous(),
                                                        8, d0, dstep, bev_w)
                finally:
                    del os.environ['FBBEV_DA_FUSED_HW']
                assert code == 0, code
                both.append(slots)
            assert torch.equal(both[0], both[1]), seed                        # the same per-wave arithmetic: the same bits
            assert not torch.isnan(slots).any(), seed
            if with_pos:      # fbbev_da_cross_attn_fused_ln: LayerNorm(output_proj(slots) + residual) in the same (8-head) workgroups
                import torch.nn.functional as F
                gg = torch.Generator().manual_seed(seed)
                Em = slots.shape[-1]
                w_o, b_o = torch.randn(Em, Em, generator=gg) * 0.2, torch.randn(Em, generator=gg) * 0.1
                lnw, lnb = torch.rand(Em, generator=gg) + 0.5, torch.randn(Em, generator=gg) * 0.1
                res = ex['query'].contiguous()
                code, y = E.da_cross_attn_fused(planes, ss, ls, pred, ref_cam, mask, qdepth, q, add,
                                                Pm[pre + 'sampling_offsets.weight'].contiguous(), Pm[pre + 'sampling_offsets.bias'].contiguous(),
                                                Pm[pre + 'attention_weights.weight'].contiguous(), Pm[pre + 'attention_weights.bias'].contiguous(),
                                                8, d0, dstep, bev_w, out_proj=(w_o, b_o, res, lnw, lnb, 1e-5))
                assert code == 0 and not torch.isnan(y).any()
                want = F.layer_norm(F.linear(slots, w_o, b_o) + res, (Em,), lnw, lnb, 1e-5)
                assert (y - want).abs().max().item() <= 2e-4, (seed, (y - want).abs().max().item())
            scale = exp.abs().max().item()
            assert (slots - exp).abs().max().item() <= 1e-4 * max(scale, 1.0), (seed, with_pos, (slots - exp).abs().max().item(), scale)
            assert (slots - unit).abs().max().item() <= 1e-4 * max(scale, 1.0), (seed, (slots - unit).abs().max().item())
    # a (Q, E) positional table shared by the samples: period Q
    args, exp, ex = _da_case(24, B=2, Q=4 * 8, E=80, M=8, P=8, DC=20, shapes=((6, 9),), extras=True)
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    Pm, pre = ex['Pm'], 'a.deformable_attention.'
    planes = value.permute(0, 2, 1, 3).contiguous()
    table = ex['qpos'][0].contiguous()                                                 # one table for both samples
    q2 = (ex['query'] + ex['qpos'] - table[None]).contiguous()                        # q2 + table == query + qpos
    code, slots = E.da_cross_attn_fused(planes, ss, ls, pred, ref_cam, mask, qdepth, q2, table,
                                        Pm[pre + 'sampling_offsets.weight'].contiguous(), Pm[pre + 'sampling_offsets.bias'].contiguous(),
                                        Pm[pre + 'attention_weights.weight'].contiguous(), Pm[pre + 'attention_weights.bias'].contiguous(),
                                        8, d0, dstep, 8)
    assert code == 0 and (slots - exp).abs().max().item() <= 1e-4 * max(exp.abs().max().item(), 1.0)
    # refused: a 1-token-wide level, P != 8
    code, _ = E.da_cross_attn_fused(planes, ss, ls, pred, ref_cam, mask, qdepth, q2, table,
                                    Pm[pre + 'sampling_offsets.weight'].contiguous(), Pm[pre + 'sampling_offsets.bias'].contiguous(),
                                    Pm[pre + 'attention_weights.weight'].contiguous(), Pm[pre + 'attention_weights.bias'].contiguous(),
                                    8, d0, dstep, 8, min_level_width=1)
    assert code == -3 or code < 0
    code, _ = E.da_cross_attn_fused(planes, ss, ls, pred, ref_cam, mask, qdepth, q2, table,
                                    Pm[pre + 'sampling_offsets.weight'].contiguous(), Pm[pre + 'sampling_offsets.bias'].contiguous(),
                                    Pm[pre + 'attention_weights.weight'].contiguous(), Pm[pre + 'attention_weights.bias'].contiguous(),
                                    4, d0, dstep, 8)
    assert code < 0


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_one_kernel_da_cross_attention_on_16bit_head_planes_emulated(dt):
    """Round 5: fbbev_rows_linear_x3_planes_e writes the value projection as bf16 / fp16 head planes (== the fp32 planes rounded once)
    and fbbev_da_cross_attn_fused_e samples them -- staged levels from LDS, the others from global memory -- with fp32 products
    and sums: EXACTLY the fp32 kernel's result on the widened planes (widening is exact, the arithmetic is the same sequence), and
    within the storage rounding of the fp32-token result (DA_SpatialCrossAttention.value_dtype; the reference keeps fp32)."""
    for seed, kw, bev_w in ((31, dict(B=2, Q=5 * 11, shapes=((16, 44), (8, 22))), 11),
                            (32, dict(B=1, Q=9 * 8, shapes=((5, 7), (9, 6), (3, 4), (2, 2))), 8)):
        args, exp, ex = _da_case(seed, E=80, M=8, P=8, DC=20, extras=True, **kw)
        value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        BN, S_, M, Dh = value.shape
        Pm, pre = ex['Pm'], 'a.deformable_attention.'
        x = ex['key'].permute(2, 0, 1, 3).reshape(BN * S_, M * Dh).contiguous()
        w_v, b_v = Pm[pre + 'value_proj.weight'].contiguous(), Pm[pre + 'value_proj.bias'].contiguous()
        code, p32 = E.rows_linear_x3_planes(x, w_v, b_v, S_, M, Dh)
        assert code == 0
        code, p16 = E.rows_linear_x3_planes(x, w_v, b_v, S_, M, Dh, dtype=dt)
        assert code == 0 and torch.equal(p16, p32.to(dt))                        # the fp32 projection rounded once (nearest even)
        q, add = ex['query'].contiguous(), ex['qpos'].reshape(-1, M * Dh).contiguous()
        w = [Pm[pre + n].contiguous() for n in ('sampling_offsets.weight', 'sampling_offsets.bias', 'attention_weights.weight',
                                                'attention_weights.bias')]
        code, s16 = E.da_cross_attn_fused(p16, ss, ls, pred, ref_cam, mask, qdepth, q, add, *w, 8, d0, dstep, bev_w)
        assert code == 0 and not torch.isnan(s16).any()
        code, swide = E.da_cross_attn_fused(p16.float().contiguous(), ss, ls, pred, ref_cam, mask, qdepth, q, add, *w, 8, d0, dstep, bev_w)
        assert code == 0 and torch.equal(s16, swide)
        code, s32 = E.da_cross_attn_fused(p32, ss, ls, pred, ref_cam, mask, qdepth, q, add, *w, 8, d0, dstep, bev_w)
        tol = (2e-2 if dt == torch.bfloat16 else 3e-3) * max(1.0, s32.abs().max().item())
        assert (s16 - s32).abs().max().item() <= tol, (s16 - s32).abs().max().item()
    code, _ = E.lib().fbbev_rows_linear_x3_planes_e(None, 0, None, None, 0, 80, 80, 10, 10, 3, None, None), None
    assert code < 0                                                               # unknown element type


@pytest.mark.parametrize('B,bh,bw,with_pos', [(1, 8, 8, True), (2, 5, 11, True), (1, 9, 16, False)])
def test_fused_bev_self_attention_emulated(B, bh, bw, with_pos):
    """fbbev_msda_self_fused -> k_msda_self_fused: mmcv MultiScaleDeformableAttention.forward as the encoder layer calls it
    (one level = the BEV grid, 4 points, value = the query tokens) from the query rows in one kernel.  Against fbbev_msda_fwd on
    the location tensor / softmaxed weights torch builds in fp32 (the in-kernel projections are split-operand bf16: ~1e-5
    relative), full and partial 8 x 8 patches, with the positional rows as addend or pre-added."""
    import torch.nn.functional as F
    M, Dh, P = 8, 10, 4
    Em, Q = M * Dh, bh * bw
    g = torch.Generator().manual_seed(Q + B)
    query = torch.randn(B, Q, Em, generator=g)
    pos = torch.randn(Q, Em, generator=g) * 0.5
    w_v, b_v = torch.randn(Em, Em, generator=g) * 0.2, torch.randn(Em, generator=g) * 0.1
    w_so, b_so = torch.randn(M * P * 2, Em, generator=g) * 0.15, torch.randn(M * P * 2, generator=g) * 2.0
    w_aw, b_aw = torch.randn(M * P, Em, generator=g) * 0.2, torch.randn(M * P, generator=g)
    xs, ys = (torch.arange(bw) + 0.5) / bw, (torch.arange(bh) + 0.5) / bh
    ref = torch.stack([xs[None].expand(bh, bw), ys[:, None].expand(bh, bw)], -1).reshape(1, Q, 1, 2).expand(B, Q, 1, 2).contiguous()
    ss, ls = torch.tensor([[bh, bw]]), torch.tensor([0])
    value = F.linear(query, w_v, b_v).view(B, Q, M, Dh).contiguous()              # value = the tokens WITHOUT the positional rows
    qp = query + pos[None]
    so = F.linear(qp, w_so, b_so).view(B, Q, M, 1, P, 2)
    aw = F.linear(qp, w_aw, b_aw).view(B, Q, M, P).softmax(-1).view(B, Q, M, 1, P).contiguous()
    norm = torch.stack([ss[..., 1], ss[..., 0]], -1)
    loc = (ref[:, :, None, :, None, :] + so / norm[None, None, None, :, None, :]).contiguous()
    base = E.msda_fwd(value, ss, ls, loc, aw)
    planes = value.permute(0, 2, 1, 3).contiguous()
    q_in, add = (query.contiguous(), pos.contiguous()) if with_pos else (qp.contiguous(), None)
    code, out = E.msda_self_fused(planes, ref, q_in, add, w_so, b_so, w_aw, b_aw, P, bw, (bh, bw))
    assert code == 0 and not torch.isnan(out).any()
    scale = max(base.abs().max().item(), 1.0)
    assert (out - base).abs().max().item() <= 1e-4 * scale, ((out - base).abs().max().item(), scale)
    # refused: two levels, 8 points
    code, _ = E.msda_self_fused(planes, ref, q_in, add, w_so, b_so, w_aw, b_aw, 8, bw, (bh, bw))
    assert code < 0
    # fbbev_msda_self_fused_ln: the block's tail in the same workgroups -- LayerNorm(output_proj(attention) + residual) -- against the
    # fp32 expression on the one-kernel attention output (split-operand output_proj: ~1e-5 relative), with and without a residual
    w_o, b_o = torch.randn(Em, Em, generator=g) * 0.2, torch.randn(Em, generator=g) * 0.1
    lnw, lnb = torch.rand(Em, generator=g) + 0.5, torch.randn(Em, generator=g) * 0.1
    for res in (query.contiguous(), None):
        code, y = E.msda_self_fused(planes, ref, q_in, add, w_so, b_so, w_aw, b_aw, P, bw, (bh, bw), out_proj=(w_o, b_o, res, lnw, lnb, 1e-5))
        assert code == 0 and not torch.isnan(y).any()
        pre = F.linear(out, w_o, b_o) + (res if res is not None else 0)
        want = F.layer_norm(pre, (Em,), lnw, lnb, 1e-5)
        assert (y - want).abs().max().item() <= 2e-4, (y - want).abs().max().item()


def test_pool_dense_eight_point_gather_batches_emulated():
    """FBBEV_POOL_GATHER8 (0x8000000): eight points per gather batch, then one batch of four, then single points -- the in-order fmaf
    chain of the default kernel (== the C oracle) for intervals of every length class (SMALL: 1 .. 30 points per voxel), with the
    re-add epilogue; refused together with 16-bit storage."""
    cfg = S.CONFIGS['SMALL']
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    B = 2
    cam = S.camera_rig(cfg, B, seed=2, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, B, seed=2)
    coor = vt.get_lidar_coor(*cam).contiguous()
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    assert int(ln.max()) >= 12 and int((ln >= 8).sum()) > 10 and int(((ln >= 4) & (ln < 8)).sum()) > 10
    Bz, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    ir = rb[st.long()].contiguous()
    counts = torch.tensor([rb.numel(), st.numel()], dtype=torch.int32)
    exp = O.bev_pool_v2(depth, feat, rd, rf, rb, (B, Z, Y, X, C), st, ln, use_fma=True)
    for tv, flags in ((64, 0x20414), (128, 0x24424), (64, 0x20400)):
        code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags | 0x8000000)
        assert code == 0 and torch.equal(out, exp), (tv, hex(flags))
    addend = torch.randn(B, C, Y, X, generator=torch.Generator().manual_seed(4))
    code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, 64, 0x20414 | 0x8000000, addend=addend)
    assert code == 0 and torch.equal(out, exp + addend[:, :, None])
    code, _ = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, 256, 0x20414 | 0x8000000)
    assert code < 0                                                        # 256-voxel tiles: no such instantiation


def test_pool_dense_pipelined_over_tile_runs_emulated():
    """FBBEV_POOL_PIPE (0x4000000) -> k_pool_fwd_dense_pipe: a workgroup walks a run of consecutive tiles, the next tile's interval
    metadata / point indices in flight under the current tile's gathers, the LDS tile re-zeroed by the store phase.  The same bits
    as the one-tile-per-workgroup kernel (== the C oracle) for runs that contain empty tiles, partial last tiles (YX no multiple of
    the tile), plane boundaries inside a run, more points per tile than the staged indices, 1 / 2 / 4 tiles per workgroup, and with
    the re-add epilogue."""
    import os
    vt = O.ViewTransformerOracle(S.CONFIGS['SMALL'].grid_config, S.CONFIGS['SMALL'].input_size, S.CONFIGS['SMALL'].downsample)
    cfg = S.CONFIGS['SMALL']
    B = 2
    cam = S.camera_rig(cfg, B, seed=1, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, B, seed=1)
    coor = vt.get_lidar_coor(*cam).contiguous()
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    Bz, Z, Y, X, C = vt.bev_feat_shape(B, cfg.channels)
    feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
    ir = rb[st.long()].contiguous()
    counts = torch.tensor([rb.numel(), st.numel()], dtype=torch.int32)
    exp = O.bev_pool_v2(depth, feat, rd, rf, rb, (B, Z, Y, X, C), st, ln, use_fma=True)
    for tv, flags in ((64, 0x20414), (128, 0x24424), (64, 0x20404)):
        code, base = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags)
        assert code == 0 and torch.equal(base, exp)
        for tpw in ('1', '2', '4'):
            os.environ['FBBEV_POOL_PIPE_TPW'] = tpw
            try:
                code, out = E.pool_dense(depth, feat, rd, rf, ir, st, ln, counts, st.numel(), B, C, Z, Y, X, tv, flags | 0x4000000)
            finally:
                del os.environ['FBBEV_POOL_PIPE_TPW']
            assert code == 0 and not torch.isnan(out).any()
            assert torch.equal(out, exp), (tv, hex(flags), tpw, (out - exp).abs().max())


@pytest.mark.parametrize('rows,I,O,with_res', [(200, 80, 80, True), (130, 320, 80, True), (70, 80, 64, False), (33, 16, 20, True)])
def test_rows_linear_layernorm_epilogue_emulated(rows, I, O, with_res):
    """fbbev_rows_linear_x3_ln: LayerNorm(x W^T + b [+ residual]) in the GEMM's store epilogue == torch's layer_norm of the fp32
    linear (+ residual) within the split-operand arithmetic (~1e-5 relative), rows that do not fill the last tile, an output width
    that is no multiple of 16, K in three chunks (the FFN's 320 -> 80)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(rows + O)
    x = torch.randn(rows, I, generator=g)
    w = torch.randn(O, I, generator=g) / I ** 0.5
    b = torch.randn(O, generator=g) * 0.3
    res = torch.randn(rows, O, generator=g) if with_res else None
    lw, lb = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2
    code, out = E.rows_linear_x3_ln(x, w, b, res, lw, lb, 1e-5)
    assert code == 0 and not torch.isnan(out).any()
    y = F.linear(x, w, b)
    ref = F.layer_norm(y + res if with_res else y, (O,), lw, lb, 1e-5)
    assert (out - ref).abs().max().item() <= 5e-5 * max(1.0, ref.abs().max().item()), (out - ref).abs().max().item()
    code, _ = E.rows_linear_x3_ln(x, torch.randn(256, I, generator=g), None, None, torch.ones(256), torch.zeros(256), 1e-5)
    assert code < 0                                                       # wider than one workgroup's output rows: refused


@pytest.mark.parametrize('rows,I,H,O,ln,with_res', [(200, 80, 320, 80, True, True), (70, 80, 64, 80, False, True), (33, 64, 128, 48, True, False),
                                                    (40, 80, 1088, 80, True, True)])      # H > 1024: b1's LDS copy takes its tail loop
def test_rows_ffn_one_kernel_emulated(rows, I, H, O, ln, with_res):
    """fbbev_rows_ffn_x3: [LayerNorm](W2 relu(W1 x + b1) + b2 [+ residual]) with the hidden rows kept in LDS fragments == the fp32
    composition in torch within the split-operand arithmetic; rows that do not fill the last tile, one / several hidden chunks."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(rows + H)
    x = torch.randn(rows, I, generator=g)
    w1, b1 = torch.randn(H, I, generator=g) / I ** 0.5, torch.randn(H, generator=g) * 0.3
    w2, b2 = torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g) * 0.3
    res = torch.randn(rows, O, generator=g) if with_res else None
    lw, lb = (torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2) if ln else (None, None)
    y = F.linear(torch.relu(F.linear(x, w1, b1)), w2, b2)
    if with_res:
        y = y + res
    ref = F.layer_norm(y, (O,), lw, lb, 1e-5) if ln else y
    import os
    for hc in ('32', '64'):                                                # hidden units per chunk: both instantiations
        os.environ['FBBEV_FFN_HC'] = hc
        try:
            code, out = E.rows_ffn_x3(x, w1, b1, w2, b2, res, lw, lb, 1e-5)
        finally:
            del os.environ['FBBEV_FFN_HC']
        assert code == 0 and not torch.isnan(out).any()
        assert (out - ref).abs().max().item() <= 5e-5 * max(1.0, ref.abs().max().item()), (hc, (out - ref).abs().max().item())
    code, _ = E.rows_ffn_x3(x, torch.randn(100, I, generator=g), torch.zeros(100), torch.randn(O, 100, generator=g), b2)
    assert code < 0                                                        # hidden width no multiple of 64: refused


@pytest.mark.parametrize('rows,Em,H,with_res', [(200, 80, 320, True), (70, 80, 64, True), (33, 64, 128, False), (130, 16, 64, True), (40, 80, 1088, True)])
def test_rows_tail_ffn_one_kernel_emulated(rows, Em, H, with_res):
    """fbbev_rows_tail_ffn_x3: LayerNorm1(y1 + W2 relu(W1 y1 + b1) + b2) with y1 = LayerNorm0(x W0^T + b0 [+ res0]) -- the
    cross-attention block's tail and the FFN block of the encoder layer (bevformer_encoder.py:250-377) in one kernel, y1 kept in
    registers and re-laid out through LDS -- against the fp32 composition in torch, and against the two kernels it replaces
    (fbbev_rows_linear_x3_ln -> fbbev_rows_ffn_x3: the same split-operand arithmetic per GEMM); rows that do not fill the last
    tile, E = 16 / 64 / 80 (one to three k-steps, a half-empty last k-step), both hidden-chunk sizes; refused shapes."""
    import os
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(rows + H + Em)
    x = torch.randn(rows, Em, generator=g)
    w0, b0 = torch.randn(Em, Em, generator=g) / Em ** 0.5, torch.randn(Em, generator=g) * 0.3
    res0 = torch.randn(rows, Em, generator=g) if with_res else None
    l0w, l0b = torch.rand(Em, generator=g) + 0.5, torch.randn(Em, generator=g) * 0.2
    w1, b1 = torch.randn(H, Em, generator=g) / Em ** 0.5, torch.randn(H, generator=g) * 0.3
    w2, b2 = torch.randn(Em, H, generator=g) / H ** 0.5, torch.randn(Em, generator=g) * 0.3
    l1w, l1b = torch.rand(Em, generator=g) + 0.5, torch.randn(Em, generator=g) * 0.2
    y0 = F.linear(x, w0, b0)
    y1 = F.layer_norm(y0 + res0 if with_res else y0, (Em,), l0w, l0b, 1e-5)
    ref = F.layer_norm(y1 + F.linear(torch.relu(F.linear(y1, w1, b1)), w2, b2), (Em,), l1w, l1b, 1e-6)
    code, t1 = E.rows_linear_x3_ln(x, w0, b0, res0, l0w, l0b, 1e-5)
    assert code == 0
    code, two = E.rows_ffn_x3(t1, w1, b1, w2, b2, residual=t1, ln_w=l1w, ln_b=l1b, eps=1e-6)
    assert code == 0
    for hc in ('32',):        # (the 64-unit chunk form of the tail kernel measured no gain and left with round 5's register-held staging)
        code, out = E.rows_tail_ffn_x3(x, w0, b0, res0, l0w, l0b, 1e-5, w1, b1, w2, b2, l1w, l1b, 1e-6)
        assert code == 0 and not torch.isnan(out).any()
        assert (out - ref).abs().max().item() <= 5e-5 * max(1.0, ref.abs().max().item()), (hc, (out - ref).abs().max().item())
        assert (out - two).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), (hc, (out - two).abs().max().item())
    # planes output (fbbev_rows_tail_ffn_x3_planes): the same bits, (images, E, tokens) instead of rows
    S = {200: 50, 70: 35, 33: 11, 130: 65, 40: 40}[rows]
    code, pl = E.rows_tail_ffn_x3(x, w0, b0, res0, l0w, l0b, 1e-5, w1, b1, w2, b2, l1w, l1b, 1e-6, tokens_per_image=S)
    assert code == 0 and torch.equal(pl, out.view(rows // S, S, Em).transpose(1, 2))
    code, _ = E.rows_tail_ffn_x3(x, w0, b0, res0, l0w, l0b, 1e-5, w1, b1, w2, b2, l1w, l1b, 1e-6, tokens_per_image=rows - 1 if rows > 1 else 2)
    assert code < 0                                                        # rows do not cover whole images
    code, _ = E.rows_tail_ffn_x3(x[:, :Em - 8].contiguous(), w0[:Em - 8, :Em - 8].contiguous(), b0[:Em - 8].contiguous(), None, l0w[:Em - 8].contiguous(),
                                  l0b[:Em - 8].contiguous(), 1e-5, w1[:, :Em - 8].contiguous(), b1, w2[:Em - 8].contiguous(), b2[:Em - 8].contiguous(),
                                  l1w[:Em - 8].contiguous(), l1b[:Em - 8].contiguous(), 1e-6)
    assert code < 0                                                        # embed no multiple of 16: refused


def test_token_pyramid_in_one_launch_emulated():
    """fbbev_tokens_from_nchw_levels == the per-level flatten(3).permute + cams_embeds + cat of bevformer.py:95-117 (bit-exact), for
    level sizes that are / are not multiples of the 32-wide transpose tile and a channel count that is not."""
    g = torch.Generator().manual_seed(7)
    n, C = 6, 80
    shapes = [(5, 9), (8, 4), (1, 3), (2, 2)]
    levels = [torch.randn(n, C, h * w, generator=g) for h, w in shapes]
    bias = torch.randn(3, C, generator=g)
    code, out = E.tokens_from_nchw_levels(levels, bias)
    assert code == 0
    exp = torch.cat([t.permute(0, 2, 1) for t in levels], 1) + bias[torch.arange(n) % 3][:, None, :]
    assert torch.equal(out, exp)
    code, out = E.tokens_from_nchw_levels(levels[:2], None)
    assert code == 0 and torch.equal(out, torch.cat([t.permute(0, 2, 1) for t in levels[:2]], 1))


def test_volume_z_reductions_emulated():
    """fbbev_volume_zreduce / _inner / fbbev_volume_z_to_front (the training path's Z-mean, the re-add's backward Z-sum, the re-layout
    of a (Y,X,Z)-contiguous gradient) against torch on the CPU emulator; refused shapes."""
    import ctypes
    g = torch.Generator().manual_seed(9)
    B, C, Z, Y, X = 2, 3, 8, 5, 12
    vol = torch.randn(B, C, Z, Y, X, generator=g)
    L = E.lib()
    out = torch.full((B, C, Y, X), float('nan'))
    assert L.fbbev_volume_zreduce(E.p(vol), B * C, Z, Y * X, ctypes.c_float(Z), E.p(out), None) == 0
    assert torch.allclose(out, vol.mean(2), rtol=1e-6, atol=1e-6)
    zl = vol.permute(0, 1, 3, 4, 2).contiguous()                                # (B, C, Y, X, Z) memory
    out2 = torch.full((B, C, Y, X), float('nan'))
    assert L.fbbev_volume_zreduce_inner(E.p(zl), B * C * Y * X, Z, ctypes.c_float(1.0), E.p(out2), None) == 0
    assert torch.allclose(out2, vol.sum(2), rtol=1e-6, atol=1e-5)
    back = torch.full((B, C, Z, Y, X), float('nan'))
    assert L.fbbev_volume_z_to_front(E.p(zl), B * C, Z, Y * X, E.p(back), None) == 0
    assert torch.equal(back, vol)
    # Y*X not a multiple of 4 / Z not a multiple of 4: refused, nothing written
    odd = torch.randn(1, 1, 6, 3, 3, generator=g)
    o = torch.full((1, 1, 3, 3), float('nan'))
    assert L.fbbev_volume_zreduce(E.p(odd), 1, 6, 9, ctypes.c_float(6.0), E.p(o), None) < 0 and torch.isnan(o).all()
    assert L.fbbev_volume_zreduce_inner(E.p(odd), 9, 6, ctypes.c_float(1.0), E.p(o), None) < 0
    assert L.fbbev_volume_z_to_front(E.p(odd), 1, 6, 9, E.p(torch.empty(1, 1, 6, 3, 3)), None) < 0


def test_owned_plane_scatter_xcd_mapping_emulated():
    """k_da_bwd_scatter_owned decodes blockIdx -> (region, sample, camera, head) in two ways: plainly, and -- when the number of
    (sample, camera) pairs divides by 8 -- so that the workgroups of one pair share an XCD.  The second form is the one the configs[2]
    training shape (B = 4, 6 cameras: 24 pairs) takes; 2 samples x 4 cameras = 8 pairs here.  Owned route (bands + whole levels, 1 and several copies) against the
    chunked route on the same inputs: the same value gradient up to the planes' scales, every word written."""
    import os
    args, exp = _da_case(41, B=2, N=4, Q=37, E=80, M=8, P=8, DC=12, shapes=((6, 9), (3, 4)))
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(41))
    shapes_host = [tuple(int(x) for x in hw) for hw in ss.tolist()]
    Dh = value.shape[-1]
    vp = torch.zeros(value.shape[:-1] + (12,)); vp[..., :Dh] = value
    got = {}
    for route, tokens in (('1', None), ('1', '20'), ('0', None)):
        os.environ['FBBEV_DA_BWD_OWNED'] = route
        os.environ['FBBEV_DA_BWD_CHUNKS'] = '2'
        os.environ['FBBEV_DA_BWD_THREADS'] = '256'
        if tokens:
            os.environ['FBBEV_DA_BWD_TOKENS'] = tokens
        try:
            got[(route, tokens)] = E.da_cross_attn_bwd(vp, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep, g, head_minor=0,
                                                       head_dim=Dh, lds_planes=True, level_hw=shapes_host)
        finally:
            for k in ('FBBEV_DA_BWD_OWNED', 'FBBEV_DA_BWD_CHUNKS', 'FBBEV_DA_BWD_THREADS', 'FBBEV_DA_BWD_TOKENS'):
                os.environ.pop(k, None)
    base = got[('0', None)]
    for key in (('1', None), ('1', '20')):
        for x, y in zip(got[key], base):
            assert not torch.isnan(x).any()
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), (key, (x - y).abs().max())
