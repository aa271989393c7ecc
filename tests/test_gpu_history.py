"""GPU parity of the temporal history fusion against fixtures recorded from the REAL
FBOCC.fuse_history (tests/golden/make_golden_history.py) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden', 'history_fusion_seq4.npz')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _module(z, dev, **kw):
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    B, C, T, Z, Y, X = (int(v) for v in z['dims'])
    m = TemporalHistoryFusion(z['dx'], z['bx'], single_bev_num_channels=C, history_cat_num=T, **kw).to(dev).eval()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    m.load_state_dict(sd)                      # the detector's own key names (fbocc.py:111-127)
    return m, (B, C, T, Z, Y, X)


def _metas(z, i):
    return [dict(sequence_group_idx=int(z[f'f{i}.seq'][b]), start_of_sequence=bool(z[f'f{i}.start'][b]),
                 curr_to_prev_ego_rt=torch.from_numpy(z[f'f{i}.ego'][b])) for b in range(len(z[f'f{i}.seq']))]


@pytest.mark.parametrize('grad', [False, True])
def test_sequence_matches_reference_fixture(dev, grad):
    z = np.load(G)
    m, (B, C, T, Z, Y, X) = _module(z, dev)
    for i in range(4):
        curr = torch.from_numpy(z[f'f{i}.curr']).to(dev).requires_grad_(grad)
        bda = torch.from_numpy(z[f'f{i}.bda']).to(dev)
        with torch.set_grad_enabled(grad):
            out = m.fuse_history(curr, _metas(z, i), bda)
        assert out.shape == (B, C, Y, X, Z)
        assert (out.detach().cpu() - torch.from_numpy(z[f'f{i}.out'])).abs().max().item() < 3e-4, i
        assert (m.history_bev.cpu() - torch.from_numpy(z[f'f{i}.history_after'])).abs().max().item() < 3e-4, i
        assert torch.equal(m.history_sweep_time, torch.from_numpy(z[f'f{i}.sweep_time_after'])), i
        if grad:
            out.sum().backward()
            assert curr.grad is not None and torch.isfinite(curr.grad).all()


def test_warp_kernel_vs_reference_grid_sample_and_oracle(dev):
    from fb_bev_amd import _capi
    from oracle import history_oracle as H
    g = torch.Generator().manual_seed(2)
    B, CH, Z, Y, X = 2, 37, 8, 50, 60
    hist = torch.randn(B, CH, Z, Y, X, generator=g)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([3.25, -1.5, 0.3])
    c, s = np.cos(0.2), np.sin(0.2)
    flow[1, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    flow[1, :3, 3] = torch.tensor([4.0, -6.0, -0.2])
    big = torch.full((B, CH + 3, Z, Y, X), float('nan'), device=dev)
    _capi.history_warp(hist.to(dev), flow.to(dev), big[:, 3:])
    grid = H.generate_grid(flow, (Z, Y, X)).permute(0, 3, 1, 2, 4)
    ref = H.grid_sample_reference(hist, grid)                   # torch's CPU grid_sample on the reference's grid
    assert (big[:, 3:].cpu() - ref).abs().max().item() < 2e-4
    assert (big[:, 3:].cpu() - H.grid_sample_3d(hist, grid)).abs().max().item() < 2e-4
    assert torch.isnan(big[:, :3]).all()


def test_history_flow_kernel(dev):
    from fb_bev_amd import _capi
    from oracle import history_oracle as H
    g = torch.Generator().manual_seed(3)
    B = 5
    def rigid():
        a = (torch.rand(1, generator=g).item() - 0.5) * 0.6
        m = torch.eye(4)
        m[:2, :2] = torch.tensor([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], dtype=torch.float32)
        m[:3, 3] = torch.randn(3, generator=g)
        return m
    hist_augs = torch.stack([rigid() for _ in range(B)]); hist_augs[:, :3, 3] = 0
    ego = torch.stack([rigid() for _ in range(B)])
    bda = torch.stack([rigid()[:3, :3] for _ in range(B)]) * torch.tensor([1., -1., 1.])
    dx, bx = torch.tensor([0.4, 0.4, 0.4]), torch.tensor([-39.8, -39.8, -0.8])
    flow = _capi.history_flow(hist_augs.to(dev), ego.to(dev), bda.contiguous().to(dev), dx.tolist(), (bx - dx / 2).tolist())
    exp = H.rt_flow(hist_augs.double(), H.forward_aug_matrix(bda.double()), ego.double(), dx.double(), bx.double())
    assert (flow.cpu().double() - exp).abs().max().item() < 2e-4        # entries up to ~200 voxels, fp32


def test_inference_path_keeps_history_as_a_view_and_is_sync_free(dev):
    z = np.load(G)
    m, (B, C, T, Z, Y, X) = _module(z, dev)
    for i in range(2):
        m.fuse_history(torch.from_numpy(z[f'f{i}.curr']).to(dev), _metas(z, i), torch.from_numpy(z[f'f{i}.bda']).to(dev))
    assert m.history_bev.data_ptr() in (m._bufs[0].data_ptr(), m._bufs[1].data_ptr())
    curr = torch.from_numpy(z['f2.curr']).to(dev)
    bda = torch.from_numpy(z['f2.bda']).to(dev)
    metas = _metas(z, 2)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        out = m.fuse_history(curr, metas, bda)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert (out.cpu() - torch.from_numpy(z['f2.out'])).abs().max().item() < 3e-4


def test_do_history_false_uses_only_the_current_frame(dev):
    z = np.load(G)
    m, (B, C, T, Z, Y, X) = _module(z, dev, do_history=False)
    outs = []
    for i in (0, 1):       # identity ego motion, different bda: rt_flow = inv(f2b).fwd.I.inv(fwd).f2b = I
        outs.append(m.fuse_history(torch.from_numpy(z['f0.curr']).to(dev), _metas(z, 0), torch.from_numpy(z[f'f{i}.bda']).to(dev)))
        assert m.history_bev is None
    assert torch.allclose(outs[0], outs[1], atol=1e-4)      # no state carried over


@pytest.mark.parametrize('B,T1,C,N,dt', [(1, 17, 80, 8000, torch.bfloat16), (2, 3, 16, 1000, torch.float32), (1, 17, 80, 4099, torch.float16)])
def test_history_conv_bf16_mfma_kernel(dev, B, T1, C, N, dt):
    """fbbev_history_conv_bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) vs float64 with the kernel's roundings restated:
    weights and frames to bf16, relu(W1 x + b1) to bf16, everything else exact.  What remains is the fp32 accumulation
    order and the occasional intermediate that rounds to the other bf16 neighbour because of it."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(C + N)
    feats = torch.randn(B, T1 * C, N, generator=g).to(dt)
    w1, w2 = torch.randn(C, C, generator=g) * 0.1, torch.randn(C, T1 * C, generator=g) * 0.05
    b1, b2 = torch.randn(B * T1, C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.2
    out = torch.full((B, C, N), float('nan'), device=dev)
    _capi.history_conv(feats.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), out, compute=torch.bfloat16)
    r = lambda t: t.bfloat16().double()  # noqa: E731
    x = r(feats.float()).view(B, T1, C, N)
    y = torch.relu(torch.einsum('oc,btcn->bton', r(w1), x) + b1.view(B, T1, C, 1).double())
    exp = torch.relu(torch.einsum('oc,bcn->bon', r(w2), r(y.float()).reshape(B, T1 * C, N)) + b2.view(1, C, 1).double())
    assert not torch.isnan(out).any()
    err = (out.cpu().double() - exp).abs()
    scale = max(1.0, exp.abs().max().item())
    assert err.max().item() <= 4e-3 * scale and err.mean().item() <= 2e-4 * scale, (err.max().item(), err.mean().item(), scale)
    # against the fp32 kernel on the same inputs: the documented ~1e-2 of the volume's peak
    ref = torch.empty_like(out)
    _capi.history_conv(feats.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), ref)
    assert (out - ref).abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize('B,T1,C,Cout,N', [(1, 17, 80, 80, 8000), (2, 3, 16, 32, 1000), (1, 2, 128, 128, 77)])
def test_history_conv_mfma_kernel(dev, B, T1, C, Cout, N):
    """fbbev_history_conv (v_mfma_f32_16x16x4_f32) vs a float64 evaluation of the same two folded convolutions."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(C + N)
    feats = torch.randn(B, T1 * C, N, generator=g)
    w1, w2 = torch.randn(C, C, generator=g) * 0.1, torch.randn(Cout, T1 * C, generator=g) * 0.05
    b1, b2 = torch.randn(B * T1, C, generator=g) * 0.2, torch.randn(Cout, generator=g) * 0.2
    out = torch.full((B, Cout, N), float('nan'), device=dev)
    _capi.history_conv(feats.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), out)
    x = feats.view(B, T1, C, N).double()
    y = torch.relu(torch.einsum('oc,btcn->bton', w1.double(), x) + b1.view(B, T1, C, 1).double())
    exp = torch.relu(torch.einsum('oc,bcn->bon', w2.double(), y.reshape(B, T1 * C, N)) + b2.view(1, Cout, 1).double())
    assert not torch.isnan(out).any()
    assert (out.cpu().double() - exp).abs().max().item() < 2e-5 * max(1.0, exp.abs().max().item())


def test_fusion_with_mfma_convs_matches_library_gemm_path(dev):
    """C = Cout = 16: the module takes fbbev_history_conv; with use_mfma_convs=False it runs the two batched library
    GEMMs -- same folded weights, same sequence, same result."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    torch.manual_seed(1)
    C, T, Z, Y, X, B = 16, 3, 4, 10, 12, 2
    m = TemporalHistoryFusion([0.8, 0.8, 0.8], [-4.4, -3.6, -0.6], single_bev_num_channels=C, history_cat_num=T).to(dev).eval()
    for seq in (m.history_keyframe_time_conv, m.history_keyframe_cat_conv):
        seq[1].running_mean.uniform_(-0.2, 0.2)
        seq[1].running_var.uniform_(0.6, 1.4)
    frames = [torch.randn(B, C, Y, X, Z, device=dev) for _ in range(3)]
    bda = torch.eye(3, device=dev)[None].repeat(B, 1, 1)
    ego = torch.eye(4)
    ego[0, 3] = 0.7

    def run(use_mfma):
        m.reset()
        m.use_mfma_convs = use_mfma
        outs = []
        for i, f in enumerate(frames):
            metas = [dict(sequence_group_idx=b, start_of_sequence=(i == 0), curr_to_prev_ego_rt=ego) for b in range(B)]
            outs.append(m.fuse_history(f, metas, bda).clone())
        return outs
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() < 1e-4


def test_config_built_path_end_to_end_at_shipped_size(dev):
    """The shipped detector config (committed extraction) -> FBViewTransform + TemporalHistoryFusion, two frames of
    synthetic 6-camera input at the shipped sizes: forward projection -> backward projection -> re-add -> history."""
    import json
    from fb_bev_amd import config as C, synthetic as S
    blocks = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'fbocc_config_path_blocks.json')))
    info = blocks['fbocc-r50-cbgs_depth_16f_16x4_20e.py']
    fvt, hist = C.build_view_transformation(info['path_blocks'])
    fvt, hist = fvt.to(dev).eval(), hist.to(dev).eval()
    assert hist.do_history is False          # the shipped config trains without history ...
    hist.do_history = True                   # ... and FBOCC.forward_test switches it on (fbocc.py:481)
    cfg = S.CONFIGS['REF']
    B = 1
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=False)]
    outs = []
    with torch.no_grad():
        for i in range(2):
            depth, ctx = S.depth_and_context(cfg, B, seed=i)
            bev = fvt(cam, ctx.to(dev), depth.to(dev))
            assert bev.shape == (B, 80, 100, 100, 8)
            ego = torch.eye(4)
            ego[0, 3] = 0.8 * i
            metas = [dict(sequence_group_idx=0, start_of_sequence=(i == 0), curr_to_prev_ego_rt=ego)]
            out = hist.fuse_history(bev, metas, cam[5])
            assert out.shape == (B, 80, 100, 100, 8) and torch.isfinite(out).all()
            outs.append(out)
    assert hist._voxel_major() and hist.history_bev.shape == (B, 16, 8 * 100 * 100, 80)     # the default ring: voxel rows
    assert hist.history_as_reference().shape == (B, 16 * 80, 8, 100, 100)                    # the reference's tensor (fbocc.py:312)
    assert (outs[0] - outs[1]).abs().max().item() > 0


# ---------------------------------------------------------------- 16-bit history ring (BASELINE configs[4])
@pytest.mark.parametrize('dt,tol', [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_sequence_with_16bit_history_ring(dev, dt, tol):
    """history_dtype = fp16 (what BASELINE configs[4] names) / bf16: the ring of T+1 frames is stored in 16 bits, taps and
    both convolutions stay fp32, a frame is rounded once when it is stored.  Stated error against the fp32 fixture of the
    REAL fuse_history over the 4-frame sequence (restarts, flips, ego motion): output within `tol` of its scale, the
    stored history within one rounding of the fp32 history per re-sampling (4 frames: <= 4 half-ulps of its scale)."""
    z = np.load(G)
    m, (B, C, T, Z, Y, X) = _module(z, dev, history_dtype=dt)
    worst = 0.0
    for i in range(4):
        curr = torch.from_numpy(z[f'f{i}.curr']).to(dev)
        bda = torch.from_numpy(z[f'f{i}.bda']).to(dev)
        with torch.no_grad():
            out = m.fuse_history(curr, _metas(z, i), bda)
        ref = torch.from_numpy(z[f'f{i}.out'])
        err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, err)
        assert m.history_bev.dtype == dt and m.history_bev.element_size() == 2
        href = torch.from_numpy(z[f'f{i}.history_after'])
        herr = (m.history_bev.float().cpu() - href).abs().max().item() / href.abs().max().item()
        ulp = 2.0 ** (-11 if dt == torch.float16 else -8)
        assert herr <= 4.5 * ulp, (i, herr)
        assert torch.equal(m.history_sweep_time, torch.from_numpy(z[f'f{i}.sweep_time_after'])), i
    print(f'{dt}: fused output max rel err vs the fp32 reference fixture over the sequence = {worst:.2e}')
    assert worst <= tol


@pytest.mark.parametrize('dt,comp', [(torch.float16, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16),
                                     (torch.float16, torch.float32), (torch.float32, torch.float32)])
def test_voxel_major_ring_equals_planar_ring(dev, dt, comp):
    """ring_layout='voxel_major' ((B,T,N,C) rows; 16-byte taps; row-operand convolutions) against the planar ring of the same
    module configuration: stored history bit-identical, fused output bit-identical with the bf16-MFMA convolutions and
    equal to fp32 rounding with the fp32-MFMA ones (same products, K summed in another order), over a sequence with a
    restart, ego motion, a flipped bda -- and across a detour through the autograd path, which hands a planar fp32 history
    back to either ring."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    B, C, T, Z, Y, X = 2, 16, 3, 4, 20, 24
    dx, bx = [0.5, 0.5, 1.0], [-5.75, -4.75, -1.5]
    torch.manual_seed(3)
    mods = [TemporalHistoryFusion(dx, bx, single_bev_num_channels=C, history_cat_num=T, history_dtype=dt,
                                  history_compute=comp, ring_layout=lay).to(dev).eval() for lay in ('planar', 'voxel_major')]
    with torch.no_grad():
        for seq in (mods[0].history_keyframe_time_conv, mods[0].history_keyframe_cat_conv):
            seq[1].running_mean.normal_(0, 0.1); seq[1].running_var.uniform_(0.5, 1.5)
    mods[1].load_state_dict(mods[0].state_dict())
    assert mods[1]._voxel_major() and not mods[0]._voxel_major()
    g = torch.Generator().manual_seed(5)
    bits = torch.int32 if dt == torch.float32 else torch.int16
    seqs = [[0, 1], [0, 1], [0, 7], [0, 7], [0, 7], [0, 7]]
    starts = [[True, True], [False, False], [False, True], [False, False], [False, False], [False, False]]
    for i in range(6):
        curr = torch.randn(B, C, Y, X, Z, generator=g)
        # half subnormals, the largest half, overflow to inf, a tie, values below the smallest subnormal: the ring's hardware
        # conversions against the planar kernels' integer rounding
        curr[0, 0, 0, :6, 0] = torch.tensor([6.0e-5, 5.96e-8, 2.0e-8, 65504.0, -1.00048828125, 3.0e-39])
        curr = curr.to(dev)
        ego = torch.eye(4).repeat(B, 1, 1)
        ego[:, 0, 3] = torch.tensor([0.4 * i, -0.3]); ego[1, :2, :2] = torch.tensor([[0.98, -0.199], [0.199, 0.98]])
        bda = torch.eye(3).repeat(B, 1, 1)
        if i >= 3:
            bda[0, 1, 1] = -1.0
        metas = [dict(sequence_group_idx=seqs[i][b], start_of_sequence=starts[i][b], curr_to_prev_ego_rt=ego[b]) for b in range(B)]
        outs = []
        for m in mods:
            if i == 4:                                                   # one frame through the autograd path
                m.train()
                outs.append(m.fuse_history(curr.clone().requires_grad_(True), metas, bda.to(dev)).detach())
                m.eval()
            else:
                with torch.no_grad():
                    outs.append(m.fuse_history(curr, metas, bda.to(dev)))
        tol = 0.0 if comp == torch.bfloat16 or i == 4 else 2e-6 * outs[0].abs().max().item()
        assert torch.allclose(outs[0], outs[1], rtol=0, atol=tol, equal_nan=True), i
        h0, h1 = mods[0].history_bev, mods[1].history_bev
        if i != 4:
            assert h0.shape == (B, T * C, Z, Y, X) and h1.shape == (B, T, Z * Y * X, C) and h1.dtype == dt
            assert torch.equal(h1.transpose(2, 3).reshape(B, T * C, Z, Y, X).contiguous().view(bits), h0.contiguous().view(bits)), i
        # (an inf tap times a zero weight is NaN in both rings, as in grid_sample)
        assert torch.allclose(mods[0].history_as_reference(), mods[1].history_as_reference(), rtol=0, atol=0, equal_nan=True), i
    assert outs[0].abs().max().item() > 0
    # overflow to inf and the ties around the largest half: the slot-0 conversion against torch's (an inf in the ring would
    # turn the sequence above into NaNs -- inf times a zero tap weight -- in either layout)
    from fb_bev_amd import _capi
    vals = torch.tensor([65504.0, 65519.0, 65520.0, 70000.0, -70000.0, 3.4e38, 1.0e-45, -0.0], device=dev)
    frame = vals.repeat(C * 8).view(1, C, 64).contiguous()
    rows = _capi.history_frame_vm(frame, torch.empty((1, 64, C), dtype=dt, device=dev))
    assert torch.equal(rows.view(bits), frame.transpose(1, 2).to(dt).contiguous().view(bits))


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('T', [3, 16])
def test_fused_warp_and_conv_kernel_equals_the_two_kernel_path(dev, dt, T):
    """fbbev_history_fused_vm (round 3: warp of the previous ring, the new ring and both bf16-MFMA convolutions in ONE launch --
    producer waves fill an LDS operand tile, consumer waves run the GEMMs from it) against fbbev_history_warp_vm +
    fbbev_history_conv_bf16 through the module (`fused_warp_conv` on / off) over a sequence with a restart, ego motion and a
    flipped bda: the stored ring AND the fused volume are the same bits (same taps, weights, roundings, operands and
    accumulation order).  X = 70: a full 64-voxel tile and a 6-voxel one per grid row."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    B, C, Z, Y, X = 2, 80, 4, 12, 70
    dx, bx = [0.5, 0.5, 1.0], [-17.25, -2.75, -1.5]
    torch.manual_seed(3)
    mods = [TemporalHistoryFusion(dx, bx, single_bev_num_channels=C, history_cat_num=T, history_dtype=dt,
                                  history_compute=torch.bfloat16, ring_layout='voxel_major').to(dev).eval() for _ in range(2)]
    with torch.no_grad():
        for seq in (mods[0].history_keyframe_time_conv, mods[0].history_keyframe_cat_conv):
            seq[1].running_mean.normal_(0, 0.1); seq[1].running_var.uniform_(0.5, 1.5)
    mods[1].load_state_dict(mods[0].state_dict())
    mods[0].fused_warp_conv, mods[1].fused_warp_conv = False, True
    assert all(m._voxel_major() for m in mods)
    g = torch.Generator().manual_seed(5)
    starts = [[True, True], [False, False], [False, True], [False, False]]
    seqs = [[0, 1], [0, 1], [0, 7], [0, 7]]
    for i in range(4):
        curr = torch.randn(B, C, Y, X, Z, generator=g).to(dev)
        ego = torch.eye(4).repeat(B, 1, 1)
        ego[:, 0, 3] = torch.tensor([0.4 * i, -0.3]); ego[1, :2, :2] = torch.tensor([[0.98, -0.199], [0.199, 0.98]])
        bda = torch.eye(3).repeat(B, 1, 1)
        if i >= 2:
            bda[0, 1, 1] = -1.0
        metas = [dict(sequence_group_idx=seqs[i][b], start_of_sequence=starts[i][b], curr_to_prev_ego_rt=ego[b]) for b in range(B)]
        with torch.no_grad():
            outs = [m.fuse_history(curr, metas, bda.to(dev)) for m in mods]
        assert torch.equal(outs[0], outs[1]), (i, (outs[0] - outs[1]).abs().max().item())
        assert torch.equal(mods[0].history_bev.view(torch.int16), mods[1].history_bev.view(torch.int16)), i
    assert outs[0].abs().max().item() > 0


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('T', [3, 16])
def test_one_kernel_and_pipelined_steps_equal_the_two_split_operand_kernels(dev, dt, T):
    """fbbev_history_fused_x3_vm (one launch: a brick's items blended in memory order, stored to the next ring and fed to both
    split-operand convolutions through an LDS tile) and fbbev_history_step_x3_vm (the two kernels launched band of rows by band on
    two streams) against fbbev_history_warp_vm + fbbev_history_conv_bf16x3: slots 1..T of the new ring and the fused volume are the
    SAME BITS.  Bricks that overhang the grid in x (70 = 4 x 16 + 6) and y (21 = 2 x 8 + 5), translation / rotation / out-of-grid
    flows, T = 16 (the detector's ring) and T = 3."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(29)
    B, C, Z, Y, X = 3, 80, 4, 21, 70
    N = Z * Y * X
    hist = (torch.randn(B, T, N, C, generator=g) * 2).to(dt).to(dev)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([1.25, -0.5, 0.25])
    flow[1, :3, :3] = torch.tensor([[0.9, -0.4, 0.0], [0.4, 0.9, 0.0], [0.0, 0.0, 1.0]])
    flow[2, :3, 3] = torch.tensor([500.0, 0.0, 0.0])
    flow = flow.to(dev)
    curr = torch.randn(B, C, N, generator=g).to(dev)
    w1, w2 = (torch.randn(C, C, generator=g) * 0.2).to(dev), (torch.randn(C, (T + 1) * C, generator=g) * 0.1).to(dev)
    b1, b2 = torch.randn(B * (T + 1), C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    ref = torch.zeros(B, T + 1, N, C, dtype=dt, device=dev)
    _capi.history_frame_vm(curr, ref[:, 0])
    _capi.history_warp_vm(hist, flow, ref[:, 1:], (Z, Y, X))
    exp = _capi.history_conv(ref, w1, b1, w2, b2, torch.empty(B, C, N, device=dev), compute='bf16x3', voxel_major=True)
    assert torch.isfinite(exp).all() and exp.abs().max().item() > 0
    for name, step in (('one kernel', lambda n, o: _capi.history_fused_x3_vm(hist, flow, n, (Z, Y, X), w1, b1, w2, b2, o)),
                       ('pipelined', lambda n, o: _capi.history_step_x3_vm(hist, flow, n, (Z, Y, X), w1, b1, w2, b2, o, chunks=3))):
        nxt = torch.full((B, T + 1, N, C), float('nan'), dtype=dt, device=dev)
        _capi.history_frame_vm(curr, nxt[:, 0])
        got = step(nxt, torch.full((B, C, N), float('nan'), device=dev))
        torch.cuda.synchronize()
        assert torch.equal(nxt.view(torch.int16), ref.view(torch.int16)), name
        assert torch.equal(got, exp), name


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_split_operand_bf16_convolutions_are_fp32_grade(dev, dt):
    """history_compute='bf16x3' (fbbev_history_conv_bf16x3: every operand split into two bf16 terms, three MFMAs per product)
    through the module on a 16-bit voxel-major ring, against the fp32-MFMA convolutions on the same ring: the stored ring is the
    same bits (the convolutions do not touch it), the fused volume within 2e-5 of its peak over a 4-frame sequence with a
    restart, ego motion and a flipped bda -- and the plain bf16 route on the same frames is > 30x further away.
    N = 4 * 12 * 70 = 3360 voxels: 26 full 128-voxel tiles and a 32-voxel one."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    B, C, Z, Y, X, T = 2, 80, 4, 12, 70, 16
    dx, bx = [0.5, 0.5, 1.0], [-17.25, -2.75, -1.5]
    torch.manual_seed(3)
    mods = [TemporalHistoryFusion(dx, bx, single_bev_num_channels=C, history_cat_num=T, history_dtype=dt,
                                  history_compute=comp, ring_layout='voxel_major').to(dev).eval()
            for comp in (torch.float32, 'bf16x3', torch.bfloat16)]
    with torch.no_grad():
        for seq in (mods[0].history_keyframe_time_conv, mods[0].history_keyframe_cat_conv):
            seq[1].running_mean.normal_(0, 0.1); seq[1].running_var.uniform_(0.5, 1.5)
    for m in mods[1:]:
        m.load_state_dict(mods[0].state_dict())
    g = torch.Generator().manual_seed(5)
    starts = [[True, True], [False, False], [False, True], [False, False]]
    worst3 = worst1 = 0.0
    for i in range(4):
        curr = torch.randn(B, C, Y, X, Z, generator=g).to(dev)
        ego = torch.eye(4).repeat(B, 1, 1)
        ego[:, 0, 3] = torch.tensor([0.4 * i, -0.3]); ego[1, :2, :2] = torch.tensor([[0.98, -0.199], [0.199, 0.98]])
        bda = torch.eye(3).repeat(B, 1, 1)
        if i >= 2:
            bda[0, 1, 1] = -1.0
        metas = [dict(sequence_group_idx=b, start_of_sequence=starts[i][b], curr_to_prev_ego_rt=ego[b]) for b in range(B)]
        with torch.no_grad():
            o32, o3, o1 = [m.fuse_history(curr, metas, bda.to(dev)) for m in mods]
        assert torch.equal(mods[0].history_bev.view(torch.int16), mods[1].history_bev.view(torch.int16)), i
        peak = o32.abs().max().item()
        worst3 = max(worst3, (o3 - o32).abs().max().item() / peak)
        worst1 = max(worst1, (o1 - o32).abs().max().item() / peak)
    assert worst3 < 2e-5 and worst1 > 30 * worst3, (worst3, worst1)


@pytest.mark.parametrize('layout', ['voxel_major', 'planar'])
def test_baseline_config4_grid_16_frame_fp16_history(dev, layout):
    """BASELINE configs[4] (stress): 400x400x16 grid, C=80, 16-frame history in fp16 = 7 GB per sample ring slot pair
    (13 GB in fp32).  Two frames through TemporalHistoryFusion at that size: (1) sequence start -- every history slot is
    the current frame, so the fused output of a voxel is a closed form of that voxel's 80 channels: checked against
    torch on 20 000 random voxels; (2) a 0.4 m ego translation along x = exactly one voxel: the re-sampled history equals
    the stored frame shifted by one voxel (to one fp16 ulp), zero-padded at the border."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    C, T, Z, Y, X = 80, 16, 16, 400, 400
    dx, bx = [0.2, 0.2, 0.4], [-39.9, -39.9, -0.8]
    torch.manual_seed(0)
    m = TemporalHistoryFusion(dx, bx, single_bev_num_channels=C, history_cat_num=T, history_dtype=torch.float16,
                              ring_layout=layout).to(dev).eval()
    assert m._voxel_major() == (layout == 'voxel_major')

    def frames():              # the ring as (T, C, Z, Y, X) fp16 frames whatever its layout (a 6.5 GB copy for voxel rows)
        h = m.history_bev
        return h.view(T, C, Z, Y, X) if h.dim() == 5 else h[0].transpose(1, 2).reshape(T, C, Z, Y, X)
    with torch.no_grad():
        for seq in (m.history_keyframe_time_conv, m.history_keyframe_cat_conv):
            seq[1].running_mean.normal_(0, 0.1); seq[1].running_var.uniform_(0.5, 1.5)
    g = torch.Generator(device=dev).manual_seed(1)
    curr = torch.randn(1, C, Y, X, Z, generator=g, device=dev)                   # (B,C,Y,X,Z) like the view transformer's output
    bda = torch.eye(3, device=dev)[None]
    meta = lambda first, ego: [dict(sequence_group_idx=0, start_of_sequence=first, curr_to_prev_ego_rt=ego)]  # noqa: E731
    with torch.no_grad():
        out0 = m.fuse_history(curr, meta(True, torch.eye(4)), bda)
    assert out0.shape == (1, C, Y, X, Z) and m.history_bev.dtype == torch.float16
    assert m.history_bev.shape == ((1, T * C, Z, Y, X) if layout == 'planar' else (1, T, Z * Y * X, C))
    assert m.history_bev.numel() * 2 == T * C * Z * Y * X * 2
    # (1) closed form on a voxel subset: all T+1 slots hold fp16(curr), time channel tau_t = 0 at a sequence start
    idx = torch.randint(0, Z * Y * X, (20000,), generator=g, device=dev)
    x16 = curr.permute(0, 1, 4, 2, 3).reshape(C, -1)[:, idx].half().float()      # (C, n) as stored
    w1, b1 = m._folded(m.history_keyframe_time_conv)
    w2, b2 = m._folded(m.history_keyframe_cat_conv)
    y = torch.relu(w1[:, :C].double() @ x16.double() + b1.double()[:, None])     # tau = 0: the time column adds nothing
    ref = torch.relu(w2.double().view(-1, T + 1, C).sum(1) @ y + b2.double()[:, None])
    got = out0.permute(0, 1, 4, 2, 3).reshape(C, -1)[:, idx].double()
    assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5
    # (2) one-voxel ego translation along x
    ego = torch.eye(4); ego[0, 3] = dx[0]
    stored = frames().clone()                                                    # fp16 frames before the second call
    with torch.no_grad():
        out1 = m.fuse_history(curr, meta(False, ego), bda)
    assert torch.isfinite(out1).all()
    h = frames()
    # slot 0 of the new history = the current frame, slot t >= 1 = previous slot t-1 sampled at x+1 (or x-1): find the sign once
    prev = stored
    # (the flow's translation is 1 voxel up to the fp32 rounding of the matrix chain and of coordinates up to 400: a tap
    # weight of ~1e-4 remains on the neighbouring voxel -- O(1e-4) absolute on N(0,1) data -- plus one fp16 rounding)
    close = lambda a, b: bool(torch.allclose(a.float(), b.float(), rtol=2e-3, atol=1e-3))  # noqa: E731
    plus = close(h[1, :, :, :, :-1], prev[0, :, :, :, 1:])
    minus = close(h[1, :, :, :, 1:], prev[0, :, :, :, :-1])
    assert plus or minus, 'a one-voxel translation must reproduce the stored frame shifted by one voxel'
    edge = h[1, :, :, :, -1] if plus else h[1, :, :, :, 0]
    assert (edge == 0).all()                                                     # zero padding outside the grid
    torch.cuda.synchronize()
    print('configs[4] history ring: %.1f GB fp16 per sample' % (m.history_bev.numel() * 2 / 2 ** 30))


@pytest.mark.parametrize('dt', [torch.float32, torch.float16])
def test_warp_lds_staged_kernel_equals_gather_kernel(dev, dt, monkeypatch):
    """k_history_warp_lds (source box of a 4096-voxel brick staged in LDS, the default) against k_history_warp (8 global
    gathers per output): the same taps, weights and fmaf chain -> bit-identical, for a translation, ego yaw + translation,
    a bda flip, a flow that leaves the grid, a rotation too large for the box (in-kernel gather path) and a NaN flow."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(7)
    B, CH, Z, Y, X = 6, 24, 8, 100, 100
    hist = torch.randn(B, CH, Z, Y, X, generator=g).to(dt).to(dev)
    flow = torch.eye(4)[None].repeat(B, 1, 1)
    flow[0, :3, 3] = torch.tensor([2.5, -1.25, 0.5])
    c, s = np.cos(0.03), np.sin(0.03)
    flow[1, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    flow[1, :3, 3] = torch.tensor([1.7, -0.6, -0.2])
    flow[2, 0, 0] = -1.0; flow[2, 0, 3] = X - 1.0
    flow[3, :3, 3] = torch.tensor([500.0, 0.0, 0.0])
    c, s = np.cos(0.9), np.sin(0.9)
    flow[4, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    flow[4, :3, 3] = torch.tensor([60.0, -20.0, 0.0])
    flow[5, 1, 1] = float('nan')
    flow = flow.to(dev)
    monkeypatch.setenv('FBBEV_HISTORY_WARP', 'direct')
    ref = _capi.history_warp(hist, flow, torch.empty_like(hist))
    monkeypatch.setenv('FBBEV_HISTORY_WARP', 'lds')
    got = _capi.history_warp(hist, flow, torch.full_like(hist, float('nan')))
    torch.cuda.synchronize()
    it = torch.int16 if dt == torch.float16 else torch.int32
    assert torch.equal(got.view(it), ref.view(it))
    assert (got[3] == 0).all() and (got[5] == 0).all() and got[4].abs().sum() > 0
