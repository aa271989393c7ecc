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
