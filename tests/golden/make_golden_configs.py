#!/usr/bin/env python3
"""Extract the `model` block (and its view-transformation sub-blocks) of every shipped FB-OCC config with fb_bev_amd.config.load_config and
store them as JSON, so the GPU box (no /root/reference) can still check that those blocks build unchanged.
Run in the build container:  python tests/golden/make_golden_configs.py"""
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get('FBBEV_REFERENCE', '/root/reference')
sys.path.insert(0, REPO)
from fb_bev_amd import config as C  # noqa: E402


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, 'occupancy_configs', 'fb_occ', '*.py'))):
        cfg = C.load_config(path)
        out[os.path.basename(path)] = {'model_type': cfg['model']['type'], 'path_blocks': C.path_blocks(cfg['model']), 'model': cfg['model'],
                                       'grid_config': cfg['grid_config'], 'data_config_input_size': list(cfg['data_config']['input_size']),
                                       'numC_Trans': cfg['numC_Trans'], 'bev_h_': cfg.get('bev_h_'), 'bev_w_': cfg.get('bev_w_')}
    dst = os.path.join(REPO, 'tests', 'golden', 'fbocc_config_path_blocks.json')
    json.dump(out, open(dst, 'w'), indent=1, sort_keys=True)
    print('wrote', dst, list(out))


if __name__ == '__main__':
    main()
