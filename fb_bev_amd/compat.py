"""Make the reference's import sites resolve to this package.

`install()` registers
    mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext  -> fb_bev_amd.bev_pool_v2_ext
so `from . import bev_pool_v2_ext` in mmdet3d/ops/bev_pool_v2/bev_pool.py:6 binds to the HIP build,
and returns an object exposing `ms_deform_attn_forward/backward`, the two attributes
`mmcv.utils.ext_loader.load_ext('_ext', [...])` asserts on
(multi_scale_deformable_attn_function.py:18-19).
"""
import sys
import types

from . import bev_pool_v2_ext, ms_deform_attn


def ext_module():
    m = types.ModuleType('fb_bev_amd._ext')
    m.ms_deform_attn_forward = ms_deform_attn.ms_deform_attn_forward
    m.ms_deform_attn_backward = ms_deform_attn.ms_deform_attn_backward
    return m


def install(force=False):
    name = 'mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext'
    if force or name not in sys.modules:
        sys.modules[name] = bev_pool_v2_ext
        parent = sys.modules.get('mmdet3d.ops.bev_pool_v2')
        if parent is not None:
            parent.bev_pool_v2_ext = bev_pool_v2_ext
    return ext_module()
