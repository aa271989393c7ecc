"""The `backward_projection` / `forward_projection` blocks of the shipped FB-OCC config
(occupancy_configs/fb_occ/fbocc-r50-cbgs_depth_16f_16x4_20e.py:78-98,149-211), as plain dicts, so the
path can be built without mmcv's Config loader.  Values are configuration data of the reference; the
`type=` strings are the registry names preserved by fb_bev_amd."""
import copy
import json
import os


def fbocc_r50(num_levels=1, bev_h=100, bev_w=100, numC_Trans=80, input_size=(256, 704),
              grid_config=None, grid_config_bevformer=None, depth_bound=(2.0, 42.0, 0.5), downsample=16):
    grid_config = grid_config or {'x': [-40, 40, 0.8], 'y': [-40, 40, 0.8], 'z': [-1, 5.4, 0.8],
                                  'depth': list(depth_bound)}
    grid_config_bevformer = grid_config_bevformer or {'x': [-40, 40, 0.8], 'y': [-40, 40, 0.8], 'z': [-1, 5.4, 1.6]}
    point_cloud_range = [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4]
    data_config = {'input_size': tuple(input_size)}
    ffn_dim = numC_Trans * 4
    forward_projection = dict(type='LSSViewTransformerFunction3D', grid_config=grid_config,
                              input_size=data_config['input_size'], downsample=downsample)
    backward_projection = dict(
        type='BackwardProjection', bev_h=bev_h, bev_w=bev_w, in_channels=numC_Trans, out_channels=numC_Trans,
        pc_range=point_cloud_range,
        transformer=dict(
            type='BEVFormer', use_cams_embeds=False, embed_dims=numC_Trans,
            encoder=dict(
                type='bevformer_encoder', num_layers=1, pc_range=point_cloud_range,
                grid_config=grid_config_bevformer, data_config=data_config, return_intermediate=False,
                transformerlayers=dict(
                    type='BEVFormerEncoderLayer',
                    attn_cfgs=[
                        dict(type='MultiScaleDeformableAttention', embed_dims=numC_Trans, dropout=0.0, num_levels=1),
                        dict(type='DA_SpatialCrossAttention', pc_range=point_cloud_range, dbound=list(depth_bound),
                             dropout=0.0,
                             deformable_attention=dict(type='DA_MSDeformableAttention', embed_dims=numC_Trans,
                                                       num_points=8, num_levels=num_levels),
                             embed_dims=numC_Trans)],
                    ffn_cfgs=dict(type='FFN', embed_dims=numC_Trans, feedforward_channels=ffn_dim, ffn_drop=0.0,
                                  act_cfg=dict(type='ReLU', inplace=True)),
                    feedforward_channels=ffn_dim, ffn_dropout=0.0,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))),
        positional_encoding=dict(type='CustormLearnedPositionalEncoding', num_feats=numC_Trans // 2,
                                 row_num_embed=bev_h, col_num_embed=bev_w))
    return copy.deepcopy(dict(forward_projection=forward_projection, backward_projection=backward_projection,
                              grid_config=grid_config, grid_config_bevformer=grid_config_bevformer,
                              data_config=data_config, depth_bound=list(depth_bound)))


# ---- the whole `model` block of the shipped detector configs, as package data --------------------------------------------------
# bench.py --mode train / tools build the full FBOCC from it on a box without /root/reference.  The JSON is what
# `fb_bev_amd.config.load_config` returns for occupancy_configs/fb_occ/*.py (`model` key), written by `extract_model_blocks`
# (run in the build container: `python -m fb_bev_amd.configs /root/reference`); tests/test_config_loading.py checks it against
# the live reference configs when the tree is mounted, and against the test fixture otherwise.
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'fbocc_model_blocks.json')
SHIPPED = 'fbocc-r50-cbgs_depth_16f_16x4_20e.py'


def model_block(name=SHIPPED):
    """The unchanged `model` dict of a shipped FB-OCC config (type FBOCC / FBOCCTRT), by config file name."""
    with open(DATA) as f:
        blocks = json.load(f)
    if name not in blocks:
        raise KeyError(f'{name!r} is not a shipped FB-OCC config ({sorted(blocks)})')
    return copy.deepcopy(blocks[name])


def extract_model_blocks(reference_root, dst=DATA):
    import glob
    from .config import load_config
    out = {}
    for path in sorted(glob.glob(os.path.join(reference_root, 'occupancy_configs', 'fb_occ', '*.py'))):
        out[os.path.basename(path)] = load_config(path)['model']
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return out


if __name__ == '__main__':
    import sys
    print(sorted(extract_model_blocks(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')))
