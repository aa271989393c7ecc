"""Pin the backward-projection oracle against fixtures produced by the REAL reference Python
(tests/golden/make_golden.py::make_backward_projection_fixtures): the rebatch / pad / scatter /
normalise logic of DA_SpatialCrossAttention.forward (spatial_cross_attention_depth.py:136-223) and the
offset / softmax / (point, Z-anchor) interleave of DA_MSDeformableAttention.forward (:513-570)."""
import os
import sys

import numpy as np
import torch

from oracle import backward_projection_oracle as BO

G = os.path.join(os.path.dirname(__file__), 'golden')
sys.path.insert(0, G)


def test_da_spatial_cross_attention_logic_matches_reference():
    from make_golden import _inner_stub
    z = np.load(os.path.join(G, 'da_sca_stub_inner.npz'))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    P = {'x.output_proj.weight': t('w'), 'x.output_proj.bias': t('b')}
    H, W = 5, 7
    out = BO.da_spatial_cross_attention(P, 'x.', t('query'), t('key'), t('key'), t('query_pos'), t('ref_cam'),
                                        t('mask'), t('qdepth'), t('pred'), torch.tensor([[H, W]]), torch.tensor([0]),
                                        z['dbound'].tolist(), num_cams=6, inner=_inner_stub)
    assert torch.allclose(out, t('out'), atol=1e-6, rtol=1e-6)
    assert not t('mask')[3].any()          # the fixture includes a camera that sees no query


def test_da_msda_sampling_locations_match_reference_cpu_branch():
    z = np.load(os.path.join(G, 'da_msda_cpu_branch.npz'))
    P = {'a.' + k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd_')}
    ss = torch.tensor([[5, 7], [3, 4]]); ls = torch.tensor([0, 35])
    out = BO.da_msda(P, 'a.', torch.from_numpy(z['q']), torch.from_numpy(z['v']), torch.from_numpy(z['ref']), ss, ls,
                     None, None, num_heads=4, num_levels=2, num_points=8, depth_weighting=False)
    assert torch.allclose(out, torch.from_numpy(z['out']), atol=1e-6, rtol=1e-6)


def test_depth_weighting_equals_single_bin_sampling():
    """(sampled depth distribution . one-hot) == bilinear sample of the query's own depth bin: the
    identity the fused HIP kernel relies on."""
    from oracle import oracle as O
    g = torch.Generator().manual_seed(0)
    B, Q, Za, DC, H, W = 3, 17, 4, 9, 6, 5
    pred = torch.rand(B, H * W, DC, generator=g)
    ref = torch.rand(B, Q, Za, 2, generator=g) * 1.2 - 0.1
    bins = torch.randint(0, DC, (B, Q, Za), generator=g)
    onehot = torch.nn.functional.one_hot(bins, DC)
    ss = torch.tensor([[H, W]])
    dref = ref.reshape(B, Q * Za, 1, 1, 1, 2)
    dsamp = O.msda_grid_sample(pred.unsqueeze(2), ss, dref, torch.ones_like(dref[..., 0])).reshape(B, Q, Za, DC)
    dw = (dsamp * onehot).sum(-1)
    single = torch.gather(dsamp, 3, bins[..., None]).squeeze(-1)
    assert torch.equal(dw, single)
