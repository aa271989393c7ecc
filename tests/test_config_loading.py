"""`occupancy_configs/fb_occ/*.py` load unchanged (north_star): the view-transformation blocks of the shipped
detector configs build through fb_bev_amd's registry names.  Runs against the live reference configs when
/root/reference is mounted (build container) and always against the committed extraction
tests/golden/fbocc_config_path_blocks.json (tests/golden/make_golden_configs.py)."""
import glob
import json
import os

import pytest

from fb_bev_amd import config as C

G = os.path.join(os.path.dirname(__file__), 'golden', 'fbocc_config_path_blocks.json')
REF = os.environ.get('FBBEV_REFERENCE', '/root/reference')
LIVE = sorted(glob.glob(os.path.join(REF, 'occupancy_configs', 'fb_occ', '*.py')))


def _check_built(fvt, hist, info):
    from fb_bev_amd.backward_projection import BackwardProjection
    from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
    fp = fvt.forward_projection
    assert isinstance(fp, LSSViewTransformerFunction3D)
    gc = info['grid_config']
    assert fp.grid_zyx == (round((gc['z'][1] - gc['z'][0]) / gc['z'][2]), round((gc['y'][1] - gc['y'][0]) / gc['y'][2]),
                           round((gc['x'][1] - gc['x'][0]) / gc['x'][2]))
    H, W = info['data_config_input_size']
    assert tuple(fp.frustum.shape) == (round((gc['depth'][1] - gc['depth'][0]) / gc['depth'][2]), H // 16, W // 16, 3)
    assert isinstance(fvt.backward_projection, BackwardProjection)
    assert fvt.backward_projection.bev_embedding.weight.shape == (info['bev_h_'] * info['bev_w_'], info['numC_Trans'])
    assert fvt.readd is True
    assert hist.history_cat_num == 16 and hist.single_bev_num_channels == info['numC_Trans']
    assert tuple(hist.history_keyframe_cat_conv[0].weight.shape)[:2] == (info['numC_Trans'], 17 * info['numC_Trans'])


def test_committed_extraction_builds():
    blocks = json.load(open(G))
    assert len(blocks) >= 2
    for name, info in blocks.items():
        fvt, hist = C.build_view_transformation(info['path_blocks'])
        _check_built(fvt, hist, info)
        dn = C.build_depth_net(info['path_blocks'])
        assert dn.depth_conv[-1].out_channels == fvt.forward_projection.frustum.shape[0]      # D depth bins
        assert dn.context_conv.out_channels == info['numC_Trans']


@pytest.mark.skipif(not LIVE, reason='reference tree not mounted (GPU box)')
@pytest.mark.parametrize('path', LIVE, ids=[os.path.basename(p) for p in LIVE])
def test_live_reference_config_loads_and_builds(path):
    cfg = C.load_config(path)
    assert cfg['model']['type'] in ('FBOCC', 'FBOCCTRT')
    assert 'dataset_type' in cfg                                  # came through _base_ / the file itself
    info = json.load(open(G))[os.path.basename(path)]
    assert json.loads(json.dumps(C.path_blocks(cfg['model']))) == info['path_blocks']    # the fixture is current
    fvt, hist = C.build_view_transformation(cfg['model'])
    _check_built(fvt, hist, info)


def test_base_merge_semantics(tmp_path):
    (tmp_path / 'base.py').write_text("a = dict(x=1, y=dict(p=1, q=2))\nb = 3\nimport os\n")
    (tmp_path / 'child.py').write_text("_base_ = ['./base.py']\na = dict(y=dict(q=5), z=7)\nc = dict(_delete_=True, k=1)\n")
    cfg = C.load_config(str(tmp_path / 'child.py'))
    assert cfg == {'a': {'x': 1, 'y': {'p': 1, 'q': 5}, 'z': 7}, 'b': 3, 'c': {'k': 1}}


def test_package_model_blocks_equal_the_reference_configs():
    """bench.py --mode train and the tools build the detector from fb_bev_amd/data/fbocc_model_blocks.json
    (fb_bev_amd.configs.model_block): it is the `model` block of the shipped configs -- the same as the test fixture's copy, and
    as the live reference configs when the tree is mounted."""
    from fb_bev_amd import configs
    fixture = json.load(open(G))
    for name, info in fixture.items():
        assert configs.model_block(name) == info['model']
    for path in LIVE:
        assert json.loads(json.dumps(C.load_config(path)['model'])) == configs.model_block(os.path.basename(path))
    with pytest.raises(KeyError):
        configs.model_block('nope.py')
