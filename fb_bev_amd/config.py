"""Load mmcv-style python config files (`occupancy_configs/fb_occ/*.py`) without mmcv, and build the
view-transformation path from the detector's `model` block unchanged.

mmcv.Config.fromfile semantics that the FB-OCC configs rely on (mmcv/utils/config.py, external): the file is
executed as python; `_base_` (str or list of str, relative to the file) names parent configs whose variables are
loaded first; dict values of the child are merged key-by-key into the parent's dicts unless the child dict carries
`_delete_=True`; everything else overrides.  Only public, non-module, non-callable top-level names are kept.

`build_view_transformation(model_cfg)` consumes exactly the keys `FBOCC.__init__` consumes for the path
(mmdet3d/models/fbbev/detectors/fbocc.py:47-131): forward_projection, backward_projection, readd, do_history,
history_cat_num, history_cat_conv_out_channels, single_bev_num_channels, interpolation_mode; `build_depth_net` consumes
the `depth_net` block (CM_DepthNet: vendor-library convolutions with the path's execution setup); `build_detector`
assembles the whole FBOCC (fb_bev_amd/fbocc.py) from the unchanged `model` block.
"""
import os
import types


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        elif isinstance(v, dict):
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
        else:
            out[k] = v
    return out


def load_config(path):
    path = os.path.abspath(path)
    scope = {'__file__': path, '__name__': '_fbbev_config_'}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), scope)    # config files are python by design (as in mmcv)
    own = {k: v for k, v in scope.items()
           if not k.startswith('__') and not isinstance(v, types.ModuleType) and not callable(v)}
    bases = own.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, load_config(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, own)


PATH_KEYS = ('depth_net', 'forward_projection', 'backward_projection', 'readd', 'do_history', 'history_cat_num',
             'history_cat_conv_out_channels', 'single_bev_num_channels', 'interpolation_mode')


def path_blocks(model_cfg):
    """The sub-dict of a detector `model` block that configures the view-transformation path."""
    return {k: model_cfg[k] for k in PATH_KEYS if k in model_cfg}


def build_view_transformation(model_cfg, with_history=True):
    """-> (FBViewTransform, TemporalHistoryFusion or None) built from the detector config, defaults of fbocc.py:47-71."""
    from .fb_view_transform import FBViewTransform
    from .history_fusion import TemporalHistoryFusion
    fvt = FBViewTransform(model_cfg['forward_projection'], model_cfg.get('backward_projection'),
                          readd=model_cfg.get('readd', False))
    hist = None
    if with_history:
        fp = fvt.forward_projection
        hist = TemporalHistoryFusion(fp.dx.tolist(), fp.bx.tolist(),
                                     single_bev_num_channels=model_cfg.get('single_bev_num_channels', 80),
                                     history_cat_num=model_cfg.get('history_cat_num', 16),
                                     history_cat_conv_out_channels=model_cfg.get('history_cat_conv_out_channels'),
                                     do_history=model_cfg.get('do_history', True),
                                     interpolation_mode=model_cfg.get('interpolation_mode', 'bilinear'))
    return fvt, hist


def build_depth_net(model_cfg, **execution_knobs):
    """-> CM_DepthNet from the detector's `depth_net` block (fbocc.py:79-80 builds it as a neck); execution_knobs:
    channels_last / compute_dtype of fb_bev_amd.depth_net.CM_DepthNet."""
    from .depth_net import CM_DepthNet
    cfg = dict(model_cfg['depth_net'])
    typ = cfg.pop('type')
    if typ != 'CM_DepthNet':
        raise KeyError(f'depth_net type {typ!r} is not part of the built path')
    cfg.update(execution_knobs)
    return CM_DepthNet(**cfg)


def build_detector(model_cfg, execution=None):
    """The whole detector from a config's `model` block (type FBOCC; the deployment config's FBOCCTRT subclass takes the same
    blocks, fbocc_trt.py) -> fb_bev_amd.fbocc.FBOCC with the reference's parameter names.  `execution`: see FBOCC."""
    from .fbocc import FBOCC
    cfg = dict(model_cfg)
    typ = cfg.pop('type', 'FBOCC')
    if typ not in ('FBOCC', 'FBOCCTRT', 'FBOCC_TRT'):
        raise KeyError(f'detector type {typ!r} is not an FB-OCC occupancy detector')
    return FBOCC(**cfg, execution=execution)
