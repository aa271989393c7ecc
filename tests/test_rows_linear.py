"""fb_bev_amd.rows_linear: the split-K backward of the backward projection's row-wise linear layers equals autograd's
(fp32 rounding), including the remainder rows, a missing bias and frozen inputs; on the CPU the module form is plain
F.linear (the custom backward is a GPU-shape optimisation)."""
import pytest
import torch
import torch.nn.functional as F

from fb_bev_amd import rows_linear as RL


@pytest.mark.parametrize('rows,slice_rows', [(1000, 64), (1003, 64), (70, 128), (4096, 2048)])
@pytest.mark.parametrize('bias', [True, False])
def test_split_k_backward_equals_autograd(rows, slice_rows, bias, monkeypatch):
    monkeypatch.setattr(RL, 'SLICE_ROWS', slice_rows)
    g = torch.Generator().manual_seed(rows + slice_rows)
    x = torch.randn(2 * rows, 24, generator=g, requires_grad=True)
    w = torch.randn(40, 24, generator=g, requires_grad=True)
    b = torch.randn(40, generator=g, requires_grad=True) if bias else None
    gy = torch.randn(2 * rows, 40, generator=g)
    y = RL._RowsLinear.apply(x, w, b)
    y.backward(gy)
    got = [x.grad.clone(), w.grad.clone()] + ([b.grad.clone()] if bias else [])
    x.grad = w.grad = None
    if bias:
        b.grad = None
    y2 = F.linear(x, w, b)
    y2.backward(gy)
    ref = [x.grad, w.grad] + ([b.grad] if bias else [])
    assert torch.equal(y, y2)
    for a, r in zip(got, ref):
        assert (a - r).abs().max() <= 2e-6 * r.abs().max()


def test_frozen_input_and_weight_get_no_gradient():
    x = torch.randn(300, 8)
    w = torch.randn(5, 8, requires_grad=True)
    y = RL._RowsLinear.apply(x, w, None)
    y.sum().backward()
    assert w.grad is not None and x.grad is None
    x2 = torch.randn(300, 8, requires_grad=True)
    y = RL._RowsLinear.apply(x2, w.detach(), None)
    y.sum().backward()
    assert x2.grad is not None


def test_module_keeps_linear_state_dict_and_cpu_route():
    m = RL.Linear(8, 5)
    ref = torch.nn.Linear(8, 5)
    ref.load_state_dict(m.state_dict())
    x = torch.randn(40000, 8, requires_grad=True)
    y = m(x)
    assert y.grad_fn is not None and 'RowsLinear' not in type(y.grad_fn).__name__     # CPU: autograd's own linear
    assert torch.equal(y, ref(x))


def test_x3_fragment_cache_follows_the_source_parameters(monkeypatch):
    """X3Weights rebuilds its fragments when -- and only when -- a SOURCE tensor changes: in-place update (_version), storage
    swap (data_ptr), a different bias; the transform (row permutation) is applied to what is cached.  (The fragment builder
    is a GPU kernel: replaced by a counter here.)"""
    calls = []
    monkeypatch.setattr(RL._capi, 'rows_linear_x3_fragments', lambda w: calls.append(w.clone()) or calls[-1])
    w = torch.nn.Parameter(torch.randn(8, 16))
    b = torch.nn.Parameter(torch.randn(8))
    perm = torch.tensor([7, 6, 5, 4, 3, 2, 1, 0])
    c = RL.X3Weights()
    c.get(w, b, lambda w_, b_: (w_[perm], b_[perm]))
    assert len(calls) == 1 and torch.equal(c.w, w.detach()[perm]) and torch.equal(c.b, b.detach()[perm])
    c.get(w, b, lambda w_, b_: (w_[perm], b_[perm]))
    assert len(calls) == 1                                            # unchanged sources: cached
    with torch.no_grad():
        w.mul_(2.0)                                                   # optimizer-style in-place update
    c.get(w, b, lambda w_, b_: (w_[perm], b_[perm]))
    assert len(calls) == 2 and torch.equal(c.w, w.detach()[perm])
    w.data = torch.randn(8, 16)                                       # storage swap (load_state_dict(assign=True), EMA): version unchanged
    c.get(w, b, lambda w_, b_: (w_[perm], b_[perm]))
    assert len(calls) == 3 and torch.equal(c.w, w.detach()[perm])
    with torch.no_grad():
        b.add_(1.0)
    c.get(w, b, lambda w_, b_: (w_[perm], b_[perm]))
    assert len(calls) == 4 and torch.equal(c.b, b.detach()[perm])
    c.get(w, None)                                                    # no bias, no transform
    assert len(calls) == 5 and c.b is None and torch.equal(c.w, w.detach())
