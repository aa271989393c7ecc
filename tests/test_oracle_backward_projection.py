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


def test_da_msda_depth_weighting_matches_reference_cuda_branch():
    """tests/golden/da_msda_cuda_branch.npz: the REAL DA_MSDeformableAttention.forward executed through its CUDA branch
    (spatial_cross_attention_depth.py:578-595) on the CPU -- is_cuda faked, the mmcv op replaced by the oracle's MSDA
    forward (tests/golden/make_golden.py).  Pins the depth weighting: distribution sampled per Z-anchor, dotted with the
    one-hot query depth, repeated over the points of an anchor, multiplied into the weights without renormalisation."""
    z = np.load(os.path.join(G, 'da_msda_cuda_branch.npz'))
    P = {'a.' + k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd_')}
    ss = torch.tensor([[5, 7], [3, 4]]); ls = torch.tensor([0, 35])
    DC = z['pred'].shape[-1]
    onehot = torch.nn.functional.one_hot(torch.from_numpy(z['bins']), DC)
    out = BO.da_msda(P, 'a.', torch.from_numpy(z['q']), torch.from_numpy(z['v']), torch.from_numpy(z['ref']), ss, ls,
                     onehot, torch.from_numpy(z['pred']), num_heads=4, num_levels=2, num_points=8, depth_weighting=True)
    ref = torch.from_numpy(z['out'])
    assert ref.abs().max() > 1e-3
    assert torch.allclose(out, ref, atol=2e-6, rtol=1e-5)
    # and it is not the CPU branch's result (the weighting changes the output)
    plain = BO.da_msda(P, 'a.', torch.from_numpy(z['q']), torch.from_numpy(z['v']), torch.from_numpy(z['ref']), ss, ls,
                       None, None, num_heads=4, num_levels=2, num_points=8, depth_weighting=False)
    assert not torch.allclose(plain, ref, atol=1e-3)


def test_self_attention_restatement_matches_the_in_tree_copy_of_mmcvs_forward():
    """mmcv's MultiScaleDeformableAttention is external, but the tree's MultiScaleDeformableAttentionTRT
    (multi_scale_deformable_attn_function.py:174-260) overrides forward with a copy of mmcv's.  The fixture is that REAL
    forward (final op = the oracle's MSDA), called the way bevformer_encoder.py:327-341 calls it."""
    z = np.load(os.path.join(G, 'mmcv_msda_forward_trt_twin.npz'))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    P = {'s.' + k[len('self_sd_'):]: t(k) for k in z.files if k.startswith('self_sd_')}
    out = BO.mmcv_msda_self_attention(P, 's.', t('self_q'), t('self_pos'), t('self_ref'), torch.tensor([[6, 5]]),
                                      torch.tensor([0]), num_heads=4, num_levels=1, num_points=4)
    assert torch.allclose(out, t('self_out'), atol=2e-6, rtol=1e-5)


def test_product_attention_module_matches_the_in_tree_copy_of_mmcvs_forward():
    """The product class (fb_bev_amd.backward_projection.MultiScaleDeformableAttention, composite path on the CPU with
    the MSDA op answered by the oracle) on both fixture cases: self-attention, and value / identity / key_padding_mask /
    batch_first=False / two levels."""
    from fb_bev_amd import backward_projection as BP
    from oracle import oracle as O
    z = np.load(os.path.join(G, 'mmcv_msda_forward_trt_twin.npz'))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731

    class OracleFn:
        @staticmethod
        def apply(value, ss, ls, loc, w, step):
            return O.msda_fwd(value.contiguous(), ss, ls, loc.contiguous(), w.contiguous())
    was = BP.MultiScaleDeformableAttnFunction_fp32
    BP.MultiScaleDeformableAttnFunction_fp32 = OracleFn
    try:
        for tag, kw, ss, ls in (('self', dict(num_levels=1, num_points=4, batch_first=True), [[6, 5]], [0]),
                                ('cross', dict(num_levels=2, num_points=3, batch_first=False), [[4, 3], [2, 2]], [0, 12])):
            m = BP.MultiScaleDeformableAttention(embed_dims=16, num_heads=4, dropout=0.0, **kw).eval()
            m.load_state_dict({k[len(tag) + 4:]: t(k) for k in z.files if k.startswith(tag + '_sd_')})
            with torch.no_grad():
                if tag == 'self':
                    out = m(t('self_q'), None, None, None, query_pos=t('self_pos'), key_pos=t('self_pos'),
                            reference_points=t('self_ref'), spatial_shapes=torch.tensor(ss), level_start_index=torch.tensor(ls))
                else:
                    out = m(t('cross_q'), None, t('cross_v'), t('cross_identity'), query_pos=t('cross_pos'),
                            key_padding_mask=t('cross_kpm'), reference_points=t('cross_ref'), spatial_shapes=torch.tensor(ss),
                            level_start_index=torch.tensor(ls))
            assert torch.allclose(out, t(tag + '_out'), atol=2e-6, rtol=1e-5), tag
    finally:
        BP.MultiScaleDeformableAttnFunction_fp32 = was


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
