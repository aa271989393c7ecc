"""The oracle is only trustworthy once pinned: check it against (a) the reference's own
known-answer test (bev_pool.py:144-175) and (b) fixtures produced by the REAL reference Python
(tests/golden/make_golden.py).  CPU-only."""
import json
import os

import numpy as np
import pytest
import torch

from fb_bev_amd import synthetic as S
from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')


def _known():
    with open(os.path.join(G, 'bev_pool_v2_known_answer.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('use_fma', [True, False])
def test_known_answer_forward_backward(use_fma):
    k = _known()
    depth = torch.tensor(k['depth']).view(*k['depth_shape'])
    feat = torch.ones(*k['feat_ones_shape'])
    rd, rf, rb = (torch.tensor(k[n]).int() for n in ('ranks_depth', 'ranks_feat', 'ranks_bev'))
    st, ln = O.intervals_from_sorted(rb)
    out = O.bev_pool_v2(depth, feat, rd, rf, rb, tuple(k['bev_feat_shape']), st, ln, use_fma)
    assert out.shape == (1, 2, 1, 2, 2)
    assert out.sum().item() == pytest.approx(k['loss'], abs=1e-6)
    gd, gf = O.bev_pool_v2_bwd(torch.ones(*k['bev_feat_shape']), depth, feat, rd, rf, rb, use_fma)
    assert torch.allclose(gd.view(-1), torch.tensor(k['grad_depth']))
    assert torch.allclose(gf.view(-1), torch.tensor(k['grad_feat']))
    # the pure-torch formulation used as cpu_baseline computes the same thing
    out_t = O.bev_pool_v2_torch(depth, feat, rd, rf, rb, tuple(k['bev_feat_shape']))
    assert torch.allclose(out_t, out, atol=1e-6)


@pytest.mark.parametrize('tag', ['TINY_B2_aug', 'SMALL_B2_aug'])
def test_index_build_matches_reference_python(tag):
    z = np.load(os.path.join(G, f'index_{tag}.npz'))
    name = tag.split('_')[0]
    cfg = S.CONFIGS[name]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
    coor = vt.get_lidar_coor(*cam)
    assert np.array_equal(coor.numpy(), z['coor'])  # same torch ops, same bits
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(torch.from_numpy(z['coor']))
    for got, key in ((rb, 'ranks_bev'), (rd, 'ranks_depth'), (rf, 'ranks_feat'),
                     (st, 'interval_starts'), (ln, 'interval_lengths')):
        assert np.array_equal(got.numpy(), z[key]), key
    depth, ctx = S.depth_and_context(cfg, 2, seed=0)
    bev = vt.view_transform(cam, depth, ctx)
    assert bev.shape == z['bev_feat'].shape
    # the fixture was summed in the reference's UNSTABLE argsort order (view_transformer.py:590),
    # the oracle in the canonical stable order: same addends, different fp32 order -> ulp-level
    assert np.allclose(bev.contiguous().numpy(), z['bev_feat'], rtol=0, atol=1e-5)


@pytest.mark.parametrize('tag', ['REF_B1', 'BL2_B1', 'BL2_B2_aug'])
def test_index_stats_full_size(tag):
    with open(os.path.join(G, 'index_stats.json')) as f:
        e = json.load(f)[tag]
    cfg = S.CONFIGS[e['config']]
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, e['B'], seed=0, bda_aug=e['bda_aug'])
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(vt.get_lidar_coor(*cam))
    assert (rb.numel(), st.numel(), int(ln.max())) == (e['P'], e['I'], e['len_max'])
    w = torch.arange(rb.numel()) % 9973 + 1
    assert int((rb.long() * w).sum()) == e['wsum']
    assert int((rd.long() * w).sum()) == e['wsum_depth']
    assert int(rf.long().sum()) == e['sum_ranks_feat']
    assert int(st.long().sum()) == e['sum_starts']


def test_point_sampling_matches_reference_python():
    z = np.load(os.path.join(G, 'point_sampling_REF_B2_aug.npz'))
    cfg = S.CONFIGS['REF']
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True)
    ref3d = O.reference_points_3d({'x': [-40, 40, 0.8], 'y': [-40, 40, 0.8], 'z': [-1, 5.4, 1.6]})
    assert np.array_equal(ref3d[:2, :2].numpy(), z['ref3d_corner'])
    ref_cam, mask, qd = O.point_sampling(ref3d, cam, (256, 704))
    sub = slice(0, 10000, 37)
    assert int(mask.sum()) == int(z['mask_count'])
    assert np.array_equal(mask[:, :, sub].numpy(), z['mask'])
    assert np.allclose(ref_cam[:, :, sub].numpy(), z['ref_cam'], rtol=0, atol=0)
    assert np.array_equal(qd[:, :, sub].numpy(), z['qdepth'])


def test_trunc_toward_zero_and_fp32_rank_quirks():
    """SURVEY section 0 traps (ii) and (i): (-1,0) voxel coords land in voxel 0; rank is fp32."""
    cfg = S.CONFIGS['TINY']
    vt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    coor = torch.full((1, 1, 1, 1, 2, 3), 0.0)
    coor[0, 0, 0, 0, 0] = torch.tensor([-8.5, -8.0, -1.0])   # x voxel coord -0.5 -> trunc 0 -> kept
    coor[0, 0, 0, 0, 1] = torch.tensor([-9.5, -8.0, -1.0])   # x voxel coord -1.5 -> -1 -> dropped
    rb, rd, rf, st, ln = vt.voxel_pooling_prepare_v2(coor)
    assert rb.tolist() == [0] and rd.tolist() == [0]
