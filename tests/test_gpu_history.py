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
    assert hist.history_bev.shape == (B, 16 * 80, 8, 100, 100)
    assert (outs[0] - outs[1]).abs().max().item() > 0
