"""Drop-in for the reference's compiled module `mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext`
(pybind11 exports at mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:104-109).

Same two names, same positional argument order -- note `interval_lengths` comes BEFORE
`interval_starts` (bev_pool.cpp:35-36,81-82) -- same in-place contract (caller pre-zeroes `out`,
`depth_grad`, `feat_grad`).  Differences by design: launches on the CURRENT HIP stream instead of
the legacy default stream, and validates dtype/device/contiguity (the reference performs no
checks and out-of-contract inputs are UB there).
"""
from . import _capi


def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                        interval_starts):
    _capi.bev_pool_v2_fwd(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                          interval_lengths)


def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                         ranks_bev, interval_lengths, interval_starts):
    _capi.bev_pool_v2_bwd(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                          ranks_bev, interval_starts, interval_lengths)
