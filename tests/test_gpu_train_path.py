"""GPU tests of the round-6 training route (fb_bev_amd/train_path.py): the encoder layer as ONE autograd node on the inference kernels,
the volume written once.  Checked against (i) the round-5 composite route (FBBEV_TRAIN_FUSED=0: round-3 kernels + fp32 vendor GEMMs under
plain autograd), itself pinned on the oracle's fp64 autograd by tests/test_gpu_backward_projection.py, and (ii) that oracle directly.
Reference: bev_pool.py:40-80, bev_pool_cuda.cu:64-118, multi_scale_deformable_attn_function.py:137-172, bevformer_encoder.py:206-377."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _graph_has(t, name):
    seen, todo = set(), [t.grad_fn]
    while todo:
        f = todo.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        if name in type(f).__name__:
            return True
        todo += [g for g, _ in f.next_functions]
    return False


def _run(fused, name, B, levels, dev, monkeypatch, seed=0):
    import train_path as T
    from fb_bev_amd import train_path as TP
    monkeypatch.setattr(TP, 'TRAIN_FUSED', fused)
    pc, m, cam, depth, ctx, mlvl = T.build(name, B, levels, dev, seed=seed)
    step, leaves, gout = T.make_step(m, cam, depth, ctx, mlvl, dev, pc, B)
    out = step()
    names = [n for n, _ in m.named_parameters()] + ['depth', 'ctx'] + [f'mlvl{i}' for i in range(1, len(mlvl or []))]
    grads = {n: (None if t.grad is None else t.grad.detach().clone()) for n, t in zip(names, leaves)}
    return out.detach(), grads, _graph_has(out, 'EncoderLayerFn'), _graph_has(out, 'PoolAdd'), step, leaves, names


def _compare(ga, gb, tag, mean_tol=1e-3, tol=5e-3, frac=0.01):
    """Every gradient tensor of route a against route b.  The routes differ by the split-operand projections (~1e-5 relative on offsets
    and logits); a sampling location within that distance of a bilinear cell boundary, or a ReLU pre-activation within it of zero,
    changes ONE query's gradient by O(its own size) -- legitimate kinks (tools/diag_bp_grad.py, test_training_paths_equal_inference_
    and_backprop use the same reading).  So: the MEAN deviation of a tensor <= mean_tol of its scale (a missing or mis-scaled term
    shows up here: observed <= 3.3e-4), at most `frac` of its entries (or two of them) beyond `tol` of the scale, none beyond a quarter of it, every entry finite."""
    worst, bad = ('', 0.0), []
    for n, a in ga.items():
        b = gb[n]
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert torch.isfinite(a).all(), n
        scale = b.abs().max().item()
        if scale == 0:
            assert a.abs().max().item() == 0, n
            continue
        err = (a - b).abs() / scale
        mean, far = err.mean().item(), (err > tol).float().mean().item()
        if mean > worst[1]:
            worst = (n, mean)
        if mean > mean_tol or far * err.numel() > max(2.0, frac * err.numel()) or err.max().item() > 0.25:
            bad.append((n, mean, far, err.max().item()))
    print(f'{tag}: worst mean gradient deviation {worst[1]:.2e} of its scale ({worst[0]})')
    assert not bad, (tag, bad)


@pytest.mark.parametrize('name,B,levels', [('REF', 2, 1), ('BL2', 1, 4)])
def test_one_node_training_route_equals_the_composite_route(dev, monkeypatch, name, B, levels):
    """forward output and EVERY gradient (parameters, depth, context, pyramid levels) of the new route against the round-5 composite
    route on the same inputs: the shipped shape (100 x 100 queries, one level, B = 2) and BASELINE configs[2] at its full size (200 x 200
    queries, 4 levels).  The two differ by the split-operand projections (~1e-5 relative) seen through bilinear slopes / ReLU kinks:
    the bars are those of _compare, observed values printed."""
    out_c, g_c, has_c, _, *_ = _run(False, name, B, levels, dev, monkeypatch)
    assert not has_c
    out_f, g_f, has_f, has_wo, *_ = _run(True, name, B, levels, dev, monkeypatch)
    assert has_f and has_wo, 'the one-node route / write-once volume was not taken'
    scale = out_c.abs().max().item()
    err = (out_f - out_c).abs().max().item()
    print(f'[{name} B={B} L={levels}] forward: max|fused - composite| = {err:.3e} on an output scale of {scale:.3f}')
    assert err <= 1e-4 * max(scale, 1.0)
    _compare(g_f, g_c, f'[{name} B={B} L={levels}] one-node vs composite')


def test_one_node_training_route_is_bit_stable_run_to_run(dev, monkeypatch):
    """the gradients the path's own kernels produce (everything but ATen's atomically reduced embedding gradients) are bit-identical
    from one step to the next on the same inputs (value-gradient bits stable run to run)"""
    _, g1, has, _, step, leaves, names = _run(True, 'BL2', 1, 4, dev, monkeypatch)
    assert has
    step()
    unstable = [n for n, t in zip(names, leaves) if t.grad is not None and not torch.equal(t.grad, g1[n])]
    print('gradients whose bits changed between two steps:', unstable)
    # `depth`: the d / d depth-distribution taps of the cross-attention are fp32 global atomics (4 taps per (query, anchor, camera), as the
    # reference's ms_deform_attn_backward accumulates its value gradient; csrc/da_bwd_planes_kernels.h:288-311) -- the value-token,
    # offset, weight and every parameter gradient must not move
    assert not [n for n in unstable if 'embed' not in n and n != 'depth'], unstable


def test_one_node_route_with_frozen_projections(dev, monkeypatch):
    """only the sampling_offsets / attention_weights heads trainable (ADVICE r3's fine-tuning case): their gradients exist and equal the
    composite route's; frozen parameters get none"""
    import train_path as T
    from fb_bev_amd import train_path as TP
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(TP, 'TRAIN_FUSED', fused)
        pc, m, cam, depth, ctx, mlvl = T.build('REF', 1, 1, dev, seed=2)
        for n, p in m.named_parameters():
            p.requires_grad_('sampling_offsets' in n or 'attention_weights' in n)
        depth.requires_grad_(False); ctx.requires_grad_(False)
        out = m(cam, ctx, depth)
        assert out.requires_grad
        g = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        out.backward(g)
        res[fused] = {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}
    _compare(res[True], res[False], 'frozen projections')
    assert sum(1 for a in res[True].values() if a is not None and a.abs().max() > 0) >= 4


def test_da_backward_on_the_forwards_head_planes_equals_the_row_entry(dev, monkeypatch):
    """fbbev_da_cross_attn_bwd_planes (the forward's head planes handed in; grad_value / grad_offsets / grad_attn allocated with
    torch.empty and WRITTEN in full) against fbbev_da_cross_attn_bwd_ws_grid on row tokens with zero-filled outputs, through the whole
    training step at the BASELINE configs[2] pyramid: every gradient bit-identical except the depth distribution's (fp32 atomics in
    both).  Unwritten words of the uninitialised outputs would show up as differences; the step is run twice per route."""
    import train_path as T
    from fb_bev_amd import train_path as TP
    res = {}
    for planes in (False, True):
        monkeypatch.setattr(TP, 'TRAIN_FUSED', True)
        monkeypatch.setattr(TP, 'DA_BWD_PLANES', planes)
        pc, m, cam, depth, ctx, mlvl = T.build('BL2', 1, 4, dev)
        step, leaves, gout = T.make_step(m, cam, depth, ctx, mlvl, dev, pc, 1)
        junk = torch.full((64 << 20,), float('nan'), device=dev)       # poison the allocator's free blocks
        del junk
        step()
        step()
        names = [n for n, _ in m.named_parameters()] + ['depth', 'ctx'] + [f'mlvl{i}' for i in range(1, len(mlvl))]
        res[planes] = {n: (None if t.grad is None else t.grad.clone()) for n, t in zip(names, leaves)}
    diff = [n for n, a in res[True].items() if a is not None and n != 'depth' and 'embed' not in n and not torch.equal(a, res[False][n])]
    assert all(torch.isfinite(a).all() for a in res[True].values() if a is not None)
    assert not diff, diff
    d = (res[True]['depth'] - res[False]['depth']).abs().max().item() / res[False]['depth'].abs().max().item()
    assert d < 1e-5, d
