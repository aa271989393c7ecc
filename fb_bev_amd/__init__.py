"""fb_bev_amd -- MI355X (gfx950) native forward-backward view transformation of FB-OCC.

Only the hot path lives here: HIP kernels behind the C ABI of
include/fbbev.h (csrc/), and the host-side mirrors of the reference's operator interface:
  bev_pool_v2_ext   <-> mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext        (compiled ext in the reference)
  bev_pool          <-> mmdet3d/ops/bev_pool_v2/bev_pool.py
  view_transformer  <-> fbbev/view_transformation/forward_projection/view_transformer.py
  ms_deform_attn    <-> mmcv._ext.ms_deform_attn_* + multi_scale_deformable_attn_function.py
There is no CPU fallback: ops raise if libfbbev_hip.so is missing or tensors are not on the GPU.
"""
__version__ = '0.1.0'
