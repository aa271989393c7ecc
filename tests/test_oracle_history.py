"""History-fusion oracle vs the fixtures produced by the REAL FBOCC.fuse_history / generate_grid
(tests/golden/make_golden_history.py; fbocc.py:169-319)."""
import os

import numpy as np
import pytest
import torch

from oracle import history_oracle as H

G = os.path.join(os.path.dirname(__file__), 'golden', 'history_fusion_seq4.npz')
G16 = os.path.join(os.path.dirname(__file__), 'golden', 'history_fusion_seq4_c16.npz')     # C = 16: MFMA / voxel-major routes


def load(path=G):
    z = np.load(path)
    B, C, T, Z, Y, X = (int(v) for v in z['dims'])
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    return z, (B, C, T, Z, Y, X), sd


def frames(z):
    i = 0
    while f'f{i}.curr' in z.files:
        yield i, {k: torch.from_numpy(z[f'f{i}.{k}']) for k in
                  ('curr', 'bda', 'ego', 'seq', 'start', 'out', 'grid', 'sampled', 'history_after', 'sweep_time_after')
                  if f'f{i}.{k}' in z.files}
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


def test_voxel_major_ring_host_logic_on_cpu_with_stubs(monkeypatch):
    """ring_layout='voxel_major' host logic (ring buffers of (B,T+1,N,C) rows, sequence restarts, the detour through the
    autograd path and back, history_as_reference) against the planar ring of the same module on CPU: every HIP entry point
    is replaced FOR THIS TEST ONLY by the oracle warp / a torch evaluation of the two folded convolutions, in the layout
    the entry point takes (the kernels themselves: tests/test_emu_kernels.py and tests/test_gpu_history.py)."""
    from fb_bev_amd import _capi
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    B, C, T, Z, Y, X = 2, 16, 3, 3, 6, 7
    dx, bx = [0.5, 0.5, 1.0], [-1.5, -1.25, -1.0]
    dxt, bxt = torch.tensor(dx), torch.tensor(bx)
    calls = {'vm': 0, 'planar': 0}

    def flow_stub(hist_augs, ego, bda, dx3, lower3):
        return H.rt_flow(hist_augs, H.forward_aug_matrix(bda), ego, dxt, bxt)

    def warp_stub(history, flow, out):
        calls['planar'] += 1
        out.copy_(H.warp_history(history.float(), flow))
        return out

    def warp_vm_stub(history, flow, out, grid_zyx):
        calls['vm'] += 1
        b, t, n, c = history.shape
        planar = history.float().transpose(2, 3).reshape(b, t * c, *grid_zyx)
        out.copy_(H.warp_history(planar, flow).reshape(b, t, c, n).transpose(2, 3))
        return out

    def frame_vm_stub(curr, out, inner=1):
        b, c, n = curr.shape
        out.copy_(curr.view(b, c, n // inner, inner).permute(0, 3, 2, 1).reshape(b, n, c))
        return out

    def conv_stub(feats, w1, bias1, w2, bias2, out, compute=torch.float32, voxel_major=False):
        assert compute == torch.bfloat16
        if voxel_major:
            b, t1, n, c = feats.shape
            x = feats.float().transpose(2, 3)
        else:
            b, tc, n = feats.shape
            c = w1.shape[0]
            t1 = tc // c
            x = feats.float().reshape(b, t1, c, n)
        y = torch.relu(torch.einsum('oc,btcn->bton', w1, x) + bias1.view(b, t1, c, 1))
        out.copy_(torch.relu(torch.einsum('oc,bcn->bon', w2, y.reshape(b, t1 * c, n)) + bias2.view(1, -1, 1)))
        return out
    for name, fn in (('history_flow', flow_stub), ('history_warp', warp_stub), ('history_warp_vm', warp_vm_stub),
                     ('history_frame_vm', frame_vm_stub), ('history_conv', conv_stub), ('require_gpu', lambda t, n: None)):
        monkeypatch.setattr(_capi, name, fn)
    torch.manual_seed(0)
    mods = [TemporalHistoryFusion(dx, bx, single_bev_num_channels=C, history_cat_num=T, history_compute=torch.bfloat16,
                                  ring_layout=lay).eval() for lay in ('planar', 'voxel_major')]
    with torch.no_grad():
        for seq in (mods[0].history_keyframe_time_conv, mods[0].history_keyframe_cat_conv):
            seq[1].running_mean.normal_(0, 0.1); seq[1].running_var.uniform_(0.5, 1.5)
    mods[1].load_state_dict(mods[0].state_dict())
    g = torch.Generator().manual_seed(1)
    seqs = [[0, 1], [0, 1], [0, 5], [0, 5], [0, 5], [0, 5]]
    starts = [[True, True], [False, False], [False, True], [False, False], [False, False], [False, False]]
    for i in range(6):
        curr = torch.randn(B, C, Y, X, Z, generator=g)
        ego = torch.eye(4).repeat(B, 1, 1)
        ego[:, 0, 3] = torch.tensor([0.3 * i, -0.2])
        bda = torch.eye(3).repeat(B, 1, 1)
        if i >= 3:
            bda[0, 1, 1] = -1.0
        metas = [dict(sequence_group_idx=seqs[i][b], start_of_sequence=starts[i][b], curr_to_prev_ego_rt=ego[b]) for b in range(B)]
        outs = []
        for m in mods:
            if i == 4:                                       # one frame through the autograd path: a planar fp32 history
                m.train()
                outs.append(m.fuse_history(curr.clone().requires_grad_(True), metas, bda).detach())
                m.eval()
            else:
                with torch.no_grad():
                    outs.append(m.fuse_history(curr, metas, bda))
        assert torch.allclose(outs[0], outs[1], atol=1e-5), i
        assert outs[0].shape == (B, C, Y, X, Z)
        if i != 4:
            assert mods[1].history_bev.shape == (B, T, Z * Y * X, C) and mods[0].history_bev.shape == (B, T * C, Z, Y, X)
            assert mods[1].history_bev.data_ptr() in (mods[1]._bufs[0].data_ptr(), mods[1]._bufs[1].data_ptr())
        assert torch.allclose(mods[0].history_as_reference(), mods[1].history_as_reference(), atol=1e-6), i
        assert torch.equal(mods[0].history_sweep_time, mods[1].history_sweep_time)
    assert calls['vm'] == 5 and calls['planar'] == 5 + 2    # five inference frames each; the autograd frame warps planar in both
    # switching the layout off mid-stream converts the ring back
    mods[1].ring_layout = 'planar'
    with torch.no_grad():
        a, b = (m.fuse_history(curr, metas, bda) for m in mods)
    assert torch.allclose(a, b, atol=1e-5) and mods[1].history_bev.shape == (B, T * C, Z, Y, X)


def test_folded_weights_cache_follows_parameter_updates():
    """_folded_pair caches the folded conv + eval-BN maps until a parameter or running statistic changes (in place, by
    load_state_dict, or by replacement)."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    m = TemporalHistoryFusion([0.5, 0.5, 1.0], [0.0, 0.0, 0.0], single_bev_num_channels=4, history_cat_num=2).eval()
    a = m._folded_pair()
    assert m._folded_pair() is a                              # unchanged: the same tuple
    w, b = m._folded(m.history_keyframe_time_conv)
    assert torch.equal(a[0], w[:, :4]) and torch.equal(a[1], w[:, 4]) and torch.equal(a[2], b)
    with torch.no_grad():
        m.history_keyframe_cat_conv[1].running_var.mul_(4.0)     # in place
    c = m._folded_pair()
    assert c is not a and torch.allclose(c[3], a[3] * (torch.sqrt(torch.tensor(1.0 + 1e-5)) / torch.sqrt(torch.tensor(4.0 + 1e-5))))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd['history_keyframe_time_conv.0.bias'] += 1.0
    m.load_state_dict(sd)
    d = m._folded_pair()
    assert d is not c and not torch.equal(d[2], c[2])
    m.history_keyframe_time_conv[0].weight = torch.nn.Parameter(m.history_keyframe_time_conv[0].weight.detach() * 2)
    assert not torch.equal(m._folded_pair()[0], d[0])


def test_oracle_reproduces_reference_sequence_c16():
    """The C = 16 sequence of the REAL fuse_history (the channel count the MFMA convolutions and the voxel-major ring take)."""
    z, (B, C, T, Z, Y, X), sd = load(G16)
    assert C == 16
    o = H.HistoryFusionOracle(H.weights_from_state_dict(sd), torch.from_numpy(z['dx']), torch.from_numpy(z['bx']), T, C)
    n = 0
    for i, f in frames(z):
        out, _, _ = o.fuse(f['curr'], f['seq'], f['start'].bool(), f['ego'], f['bda'])
        assert torch.allclose(out, f['out'], atol=2e-4), i
        assert torch.equal(o.history_sweep_time, f['sweep_time_after']), i
        if 'history_after' in f:
            assert torch.allclose(o.history_bev, f['history_after'], atol=1e-4), i
            n += 1
    assert n == 1


@pytest.mark.parametrize('layout,compute,dt,tol_out,tol_hist', [
    ('planar', torch.float32, torch.float32, 2e-5, 2e-5),                 # measured 1.3e-6 / 1.1e-6 of the scale
    ('voxel_major', torch.float32, torch.float32, 2e-5, 2e-5),            # 1.1e-6 / 1.1e-6
    ('voxel_major', torch.float32, torch.float16, 1e-3, 1.5e-3),          # 2.2e-4 / 3.6e-4
    ('planar', torch.bfloat16, torch.float32, 1e-2, 2e-5),                # 3.6e-3 / 1.1e-6
    ('voxel_major', torch.bfloat16, torch.bfloat16, 1e-2, 8e-3),          # 3.3e-3 / 3.9e-3
])
def test_product_host_code_on_emulated_kernels_vs_reference_sequence_c16(monkeypatch, layout, compute, dt, tol_out, tol_hist):
    """fb_bev_amd.history_fusion END TO END on CPU against the fixture of the REAL fuse_history: the module's host code
    drives the product kernels compiled for the CPU emulator (flow, warp, slot-0 transpose, both fused convolutions --
    fp32 MFMA / bf16 MFMA, planar / voxel-major ring, fp32 / 16-bit storage), i.e. everything but the GPU itself.
    Tolerances (relative to the tensor's largest magnitude; measured values next to each case): fp32 routes 2e-5, 16-bit
    storage a few half-ulps of the frames per re-sampling, bf16 arithmetic 1e-2 of the output scale."""
    from fb_bev_amd import _capi
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'emu'))
    import emu_capi as E
    z, (B, C, T, Z, Y, X), sd = load(G16)

    def flow_stub(hist_augs, ego, bda, dx3, lower3):
        return E.history_flow(hist_augs, ego, bda, dx3, lower3)

    def warp_stub(history, flow, out):
        return E.history_warp(history, flow, out=out)

    def warp_vm_stub(history, flow, out, grid_zyx):
        return E.history_warp_vm(history, flow, grid_zyx, out=out)

    def frame_vm_stub(curr, out, inner=1):
        return E.history_frame_vm(curr, out.dtype, out=out, inner=inner)

    def conv_stub(feats, w1, bias1, w2, bias2, out, compute=torch.float32, voxel_major=False):
        out.copy_(E.history_conv(feats, w1.contiguous(), bias1.contiguous(), w2.contiguous(), bias2.contiguous(),
                                 bf16=compute == torch.bfloat16, voxel_major=voxel_major))
        return out
    for name, fn in (('history_flow', flow_stub), ('history_warp', warp_stub), ('history_warp_vm', warp_vm_stub),
                     ('history_frame_vm', frame_vm_stub), ('history_conv', conv_stub), ('require_gpu', lambda t, n: None)):
        monkeypatch.setattr(_capi, name, fn)
    m = TemporalHistoryFusion(z['dx'], z['bx'], single_bev_num_channels=C, history_cat_num=T, history_dtype=dt,
                              history_compute=compute, ring_layout=layout).eval()
    m.load_state_dict(sd)
    assert m._voxel_major() == (layout == 'voxel_major')
    for i, f in frames(z):
        metas = [dict(sequence_group_idx=int(f['seq'][b]), start_of_sequence=bool(f['start'][b]),
                      curr_to_prev_ego_rt=f['ego'][b]) for b in range(B)]
        curr = f['curr'].clone()
        if i % 2 == 1:           # as the view transformation hands it over: a (Y, X, Z)-shaped view of a (Z, Y, X) buffer
            curr = curr.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
            assert not curr.is_contiguous()
        with torch.no_grad():
            out = m.fuse_history(curr, metas, f['bda'])
        scale = f['out'].abs().max().item()
        assert (out - f['out']).abs().max().item() <= tol_out * scale, (i, (out - f['out']).abs().max().item(), scale)
        assert torch.equal(m.history_sweep_time, f['sweep_time_after']), i
        if 'history_after' in f:
            h = m.history_as_reference()
            hs = f['history_after'].abs().max().item()
            assert (h - f['history_after']).abs().max().item() <= tol_hist * hs, (i, (h - f['history_after']).abs().max().item(), hs)
    assert m.history_bev.dim() == (4 if layout == 'voxel_major' else 5) and m.history_bev.dtype == dt
