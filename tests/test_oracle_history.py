"""History-fusion oracle vs the fixtures produced by the REAL FBOCC.fuse_history / generate_grid
(tests/golden/make_golden_history.py; fbocc.py:169-319)."""
import os

import numpy as np
import torch

from oracle import history_oracle as H

G = os.path.join(os.path.dirname(__file__), 'golden', 'history_fusion_seq4.npz')


def load():
    z = np.load(G)
    B, C, T, Z, Y, X = (int(v) for v in z['dims'])
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    return z, (B, C, T, Z, Y, X), sd


def frames(z):
    i = 0
    while f'f{i}.curr' in z.files:
        yield i, {k: torch.from_numpy(z[f'f{i}.{k}']) for k in
                  ('curr', 'bda', 'ego', 'seq', 'start', 'out', 'grid', 'sampled', 'history_after', 'sweep_time_after')}
        i += 1


def test_trilinear_restatement_matches_torch_kernel():
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(2, 5, 4, 6, 7, generator=g)
    grid = torch.rand(2, 3, 5, 6, 3, generator=g) * 2.6 - 1.3          # includes out-of-range samples (zero padding)
    assert torch.allclose(H.grid_sample_3d(inp, grid), H.grid_sample_reference(inp, grid), atol=1e-6)
    # exact lattice points reproduce the input
    Z, Y, X = inp.shape[2:]
    ident = H.generate_grid(torch.eye(4)[None].repeat(2, 1, 1), (Z, Y, X)).permute(0, 3, 1, 2, 4)
    assert torch.allclose(H.grid_sample_3d(inp, ident), inp, atol=1e-6)


def test_oracle_reproduces_reference_sequence():
    z, (B, C, T, Z, Y, X), sd = load()
    o = H.HistoryFusionOracle(H.weights_from_state_dict(sd), torch.from_numpy(z['dx']), torch.from_numpy(z['bx']), T, C)
    for i, f in frames(z):
        out, sampled, flow = o.fuse(f['curr'], f['seq'], f['start'].bool(), f['ego'], f['bda'])
        grid = H.generate_grid(flow, (Z, Y, X)).permute(0, 3, 1, 2, 4)
        assert torch.allclose(grid, f['grid'], atol=2e-5), i                 # 4x4 products: LAPACK vs restated order
        assert torch.allclose(sampled, f['sampled'], atol=1e-4), i
        assert torch.allclose(out, f['out'], atol=1e-4), i
        assert torch.allclose(o.history_bev, f['history_after'], atol=1e-4), i
        assert torch.equal(o.history_sweep_time, f['sweep_time_after']), i


def test_sequence_restart_resets_history():
    z, (B, C, T, Z, Y, X), sd = load()
    fr = dict(frames(z))
    assert fr[2]['start'].tolist() == [False, True]
    # sample 1 restarts at frame 2: its history is the current frame repeated, sweep times zero
    hist = fr[2]['history_after'][1].view(T, C, Z, Y, X)
    curr = fr[2]['curr'][1].permute(0, 3, 1, 2)
    for t in range(T):
        assert torch.allclose(hist[t], curr, atol=1e-5)
    assert fr[2]['sweep_time_after'][1].tolist() == [0.0] * T
    assert fr[2]['sweep_time_after'][0].tolist() == [0.0, 1.0, 2.0]


def test_module_state_logic_on_cpu_with_oracle_standing_in_for_the_hip_calls(monkeypatch):
    """Host logic of fb_bev_amd.history_fusion (buffers, restarts, folded convs, both paths) exercised on CPU: the two
    HIP entry points are replaced by the oracle FOR THIS TEST ONLY (the product has no such fallback; the real
    kernels are checked by tests/test_gpu_history.py and the emulator tests)."""
    from fb_bev_amd import _capi
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    z, (B, C, T, Z, Y, X), sd = load()
    dx, bx = torch.from_numpy(z['dx']), torch.from_numpy(z['bx'])

    def flow_stub(hist_augs, ego, bda, dx3, lower3):
        return H.rt_flow(hist_augs, H.forward_aug_matrix(bda), ego, dx, bx)

    def warp_stub(history, flow, out):
        out.copy_(H.warp_history(history, flow))
        return out
    monkeypatch.setattr(_capi, 'history_flow', flow_stub)
    monkeypatch.setattr(_capi, 'history_warp', warp_stub)
    monkeypatch.setattr(_capi, 'require_gpu', lambda t, name: None)
    for grad in (False, True):
        m = TemporalHistoryFusion(z['dx'], z['bx'], single_bev_num_channels=C, history_cat_num=T).eval()
        m.load_state_dict(sd)
        for i, f in frames(z):
            metas = [dict(sequence_group_idx=int(f['seq'][b]), start_of_sequence=bool(f['start'][b]),
                          curr_to_prev_ego_rt=f['ego'][b]) for b in range(B)]
            with torch.set_grad_enabled(grad):
                out = m.fuse_history(f['curr'].clone().requires_grad_(grad), metas, f['bda'])
            assert torch.allclose(out.detach(), f['out'], atol=2e-4), (grad, i)
            assert torch.allclose(m.history_bev, f['history_after'], atol=2e-4), (grad, i)
            assert torch.equal(m.history_sweep_time, f['sweep_time_after']), (grad, i)
        if not grad:
            assert m.history_bev.data_ptr() in (m._bufs[0].data_ptr(), m._bufs[1].data_ptr())   # a view, not a clone
