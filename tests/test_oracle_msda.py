"""MSDeformAttn oracle (mmcv-full 1.5.2 semantics restated, PARITY UNPINNED vs mmcv itself):
cross-check the loop-exact C im2col/col2im against the F.grid_sample formulation and autograd."""
import pytest
import torch

from oracle import oracle as O


def make_case(B, Q, M, Dh, shapes, P, seed=0, lo=-0.2, hi=1.2):
    g = torch.Generator().manual_seed(seed)
    ss = torch.tensor(shapes, dtype=torch.int64)
    ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    S = int((ss[:, 0] * ss[:, 1]).sum())
    L = len(shapes)
    value = torch.randn(B, S, M, Dh, generator=g)
    loc = torch.rand(B, Q, M, L, P, 2, generator=g) * (hi - lo) + lo   # includes out-of-range samples
    w = torch.rand(B, Q, M, L, P, generator=g)
    return value, ss, ls, loc, w


CASES = [
    dict(B=2, Q=37, M=3, Dh=10, shapes=[[6, 9], [3, 5]], P=4),      # head dim 10 like FB-OCC
    dict(B=1, Q=64, M=8, Dh=10, shapes=[[16, 44]], P=8),            # cross-attn value call shape
    dict(B=2, Q=50, M=1, Dh=59, shapes=[[8, 11]], P=1),             # depth-sampling call (1 head x D)
    dict(B=1, Q=20, M=2, Dh=4, shapes=[[4, 4], [2, 2], [1, 1]], P=2),
]


@pytest.mark.parametrize('case', CASES)
def test_msda_c_vs_grid_sample(case):
    value, ss, ls, loc, w = make_case(**case)
    out_c = O.msda_fwd(value, ss, ls, loc, w)
    out_t = O.msda_grid_sample(value, ss, loc, w)
    assert torch.allclose(out_c, out_t, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('case', CASES)
def test_msda_bwd_c_vs_autograd(case):
    value, ss, ls, loc, w = make_case(**case, seed=3)
    value_t, loc_t, w_t = (t.clone().double().requires_grad_() for t in (value, loc, w))
    out = O.msda_grid_sample(value_t, ss, loc_t, w_t)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    out.backward(go.double())
    gv, gl, gw = O.msda_bwd(value, ss, ls, loc, w, go)
    assert torch.allclose(gv.double(), value_t.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(gw.double(), w_t.grad, atol=1e-4, rtol=1e-4)
    # d/dloc is discontinuous exactly on pixel boundaries; random points are a.s. off them
    assert torch.allclose(gl.double(), loc_t.grad, atol=2e-3, rtol=1e-3)


def test_msda_edge_semantics():
    """Corner zero-padding and the (-1, size) validity window (row (a)17 of SURVEY section 8)."""
    ss = torch.tensor([[2, 2]]); ls = torch.tensor([0])
    value = torch.tensor([1., 2., 3., 4.]).view(1, 4, 1, 1)
    def at(x, y):
        loc = torch.tensor([x, y]).view(1, 1, 1, 1, 1, 2)
        return O.msda_fwd(value, ss, ls, loc, torch.ones(1, 1, 1, 1, 1)).item()
    assert at(0.25, 0.25) == pytest.approx(1.0)       # pixel centre (0,0)
    assert at(0.5, 0.5) == pytest.approx(2.5)         # mean of the four
    assert at(0.0, 0.0) == pytest.approx(0.25)        # half-outside in x and y: 1/4 weight
    assert at(-0.3, 0.5) == 0.0                       # w_im = -1.1 -> rejected
    assert at(1.0, 0.75) == pytest.approx(0.5 * 4.0)  # right edge, half weight on pixel (1,1)
