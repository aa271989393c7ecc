"""Test helper: the reference tree's twin of mmcv's ms_deform_attn bilinear device functions
(mmdet3d/ops/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-147), compiled from where it lies by `make -C oracle ref_msda`
into oracle/_ref/libmsda_bilinear_ref.so with the functions callable on the host.  Checker only."""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_void_p

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libmsda_bilinear_ref.so')
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(PATH)
        _lib.ref_im2col_bilinear.restype = c_float
        _lib.ref_im2col_bilinear.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int]
        _lib.ref_col2im_bilinear.restype = None
        _lib.ref_col2im_bilinear.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_float,
                                             c_float, c_float, c_void_p, POINTER(c_float), POINTER(c_float)]
    return _lib


def im2col(data, H, W, heads, ch, h, w, m, c):
    assert data.dtype == np.float32 and data.flags['C_CONTIGUOUS']
    return float(lib().ref_im2col_bilinear(data.ctypes.data, H, W, heads, ch, float(h), float(w), m, c))


def col2im(data, H, W, heads, ch, h, w, m, c, offset_scale, top_grad, mask, grad_im):
    """-> (grad_offset[0], grad_offset[1], grad_mask); grad_im accumulated in place"""
    go = (c_float * 2)()
    gm = (c_float * 1)()
    lib().ref_col2im_bilinear(data.ctypes.data, H, W, heads, ch, float(h), float(w), m, c, float(offset_scale), float(top_grad),
                              float(mask), grad_im.ctypes.data, go, gm)
    return float(go[0]), float(go[1]), float(gm[0])
