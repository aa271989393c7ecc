"""Test helper: call the REFERENCE's own bev_pool kernels (mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu,
compiled unmodified for gfx950 by oracle/Makefile into oracle/_ref/libbev_pool_ref.so).
Checker only -- never imported by the product."""
import ctypes
import os
from ctypes import c_int, c_void_p

import torch

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref',
                    'libbev_pool_ref.so')
# C++-mangled names of bev_pool_v2(...) / bev_pool_v2_grad(...)  (bev_pool_cuda.cu:122,130)
FWD = '_Z11bev_pool_v2iiPKfS0_PKiS2_S2_S2_S2_Pf'
BWD = '_Z16bev_pool_v2_gradiiPKfS0_S0_PKiS2_S2_S2_S2_PfS3_'
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(PATH)
        getattr(_lib, FWD).argtypes = [c_int, c_int] + [c_void_p] * 8
        getattr(_lib, FWD).restype = None
        getattr(_lib, BWD).argtypes = [c_int, c_int] + [c_void_p] * 10
        getattr(_lib, BWD).restype = None
    return _lib


def p(t):
    assert t.is_cuda and t.is_contiguous()
    return c_void_p(t.data_ptr())


def fwd(depth, feat, rd, rf, rb, starts, lengths, out):
    """reference launcher: legacy default stream, so fence both sides."""
    torch.cuda.synchronize()
    getattr(lib(), FWD)(feat.shape[-1], starts.numel(), p(depth), p(feat), p(rd), p(rf), p(rb), p(starts),
                        p(lengths), p(out))
    torch.cuda.synchronize()


def bwd(out_grad, depth, feat, rd, rf, rb, starts, lengths, depth_grad, feat_grad):
    torch.cuda.synchronize()
    getattr(lib(), BWD)(out_grad.shape[-1], starts.numel(), p(out_grad), p(depth), p(feat), p(rd), p(rf), p(rb),
                        p(starts), p(lengths), p(depth_grad), p(feat_grad))
    torch.cuda.synchronize()
