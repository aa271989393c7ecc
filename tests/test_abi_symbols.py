"""CPU-side checks of the drop-in boundary: libfbbev_hip.so loads and exports every symbol that
include/fbbev.h declares; the Python signature table matches the header; the product refuses CPU
tensors (no fallback).  No compute calls are made (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

from fb_bev_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'fbbev.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return {m.group(2): m.group(3) for m in
            re.finditer(r'\b(int|size_t)\s+(fbbev_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S)}


def test_header_and_signature_table_agree():
    decl = header_functions()
    assert set(decl) == set(_capi.SIGNATURES), set(decl) ^ set(_capi.SIGNATURES)
    for name, args in decl.items():
        n = 0 if args.strip() in ('void', '') else len(args.split(','))
        assert n == len(_capi.SIGNATURES[name][1]), name


def test_library_exports_every_symbol():
    assert os.path.exists(_capi.LIB_PATH), 'run `python -m fb_bev_amd.build` (or __graft_entry__.build())'
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name
    _capi.declare(lib)
    assert lib.fbbev_version() >= 100


def test_no_cpu_fallback():
    from fb_bev_amd import bev_pool_v2_ext, ms_deform_attn
    x = torch.zeros(1, 1, 2, 2, 2)
    i = torch.zeros(2, dtype=torch.int32)
    with pytest.raises(_capi.FbbevError):
        bev_pool_v2_ext.bev_pool_v2_forward(x, x, x.clone(), i, i, i, i, i)
    with pytest.raises(_capi.FbbevError):
        ms_deform_attn.ms_deform_attn_forward(torch.zeros(1, 4, 1, 1), torch.tensor([[2, 2]]), torch.tensor([0]),
                                              torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1))


def test_compat_install_binds_reference_names():
    import sys
    from fb_bev_amd import compat
    ext = compat.install(force=True)
    assert sys.modules['mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext'].bev_pool_v2_forward
    assert callable(ext.ms_deform_attn_forward) and callable(ext.ms_deform_attn_backward)
    del sys.modules['mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext']


def test_invalid_arguments_return_error_codes_without_touching_the_gpu():
    """Error convention of include/fbbev.h: <0 for invalid arguments, never an exception across the ABI.
    Every call below is rejected by the argument checks BEFORE any launch, so this runs without a GPU."""
    lib = _capi.declare(ctypes.CDLL(_capi.LIB_PATH))
    NULL = ctypes.c_void_p(0)
    P = ctypes.c_void_p(0x1000)          # non-null dummy; never dereferenced on these paths
    assert lib.fbbev_bev_pool_v2_fwd(0, 4, P, P, P, P, P, P, P, P, NULL) == -1          # c <= 0
    assert lib.fbbev_bev_pool_v2_fwd(80, -1, P, P, P, P, P, P, P, P, NULL) == -1        # n_intervals < 0
    assert lib.fbbev_bev_pool_v2_fwd(80, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == 0   # empty: no-op
    assert lib.fbbev_bev_pool_v2_fwd(80, 4, NULL, P, P, P, P, P, P, P, NULL) == -1      # null depth
    assert lib.fbbev_bev_pool_v2_bwd(300, 4, P, P, P, P, P, P, P, P, P, P, NULL) == -2  # c > 256 unsupported
    assert lib.fbbev_lidar_coor(P, P, P, P, P, P, P, P, P, 0, 6, 8, 4, 6, P, NULL) == -1
    assert lib.fbbev_rank_build(NULL, 1, 6, 8, 4, 6, P, P, P, P, P, P, P, P, P, P, P, 1 << 30, NULL) == -1
    assert lib.fbbev_rank_build(P, 1, 6, 8, 4, 6, P, P, P, P, P, P, P, P, P, P, P, 16, NULL) == -3      # workspace too small
    assert lib.fbbev_pool_tile_index(P, P, P, 10, 1, 16, 200, 200, 128, 0, P, 8, NULL) == -3
    assert lib.fbbev_bev_pool_v2_dense_fwd(P, P, P, P, P, P, P, 1, 6, 4, 16, 16, P, 0, 0, P, 1 << 20, 128, 0, NULL) == -2  # C % 4
    assert lib.fbbev_bev_pool_v2_dense_fwd(P, P, P, P, P, P, P, 1, 8, 4, 16, 16, P, 0, 17, P, 1 << 20, 128, 0, NULL) == -1  # bad stride
    assert lib.fbbev_msda_fwd(P, P, P, P, P, 1, 0, 8, 10, 1, 5, 4, P, NULL) == -1       # spatial_size <= 0
    assert lib.fbbev_msda_fwd(P, P, P, P, P, 0, 704, 8, 10, 1, 5, 4, P, NULL) == 0      # empty batch: no-op
    assert lib.fbbev_da_cross_attn_fwd(P, P, P, P, P, P, P, P, P, 1, 6, 704, 8, 10, 1, 100, 8, 16, 80, 2.0, 0.5, 0, 0, P, NULL) == -2  # Za > 8
    assert lib.fbbev_da_cross_attn_fwd(P, P, P, P, P, P, P, P, P, 1, 6, 704, 8, 10, 1, 100, 8, 4, 80, 2.0, 0.0, 0, 0, P, NULL) == -1   # dstep == 0
    # temporal history fusion: the reference-layout entries and the voxel-major ring
    P16, P8 = ctypes.c_void_p(0x1000), ctypes.c_void_p(0x1008)     # 16-byte aligned / not
    assert lib.fbbev_history_warp_e(P, 0, P, 1, 80, 1, 8, 8, P, 0, 0, NULL) == -1          # Z < 2 (the reference divides by Z - 1)
    assert lib.fbbev_history_warp_e(P, 0, P, 0, 80, 8, 8, 8, P, 0, 0, NULL) == 0           # empty batch: no-op
    assert lib.fbbev_history_warp_vm(P16, 0, P, 1, 16, 80, 8, 8, 8, P16, 0, 3, NULL) == -1  # elem_type
    assert lib.fbbev_history_warp_vm(P16, 0, P, 1, 16, 84, 8, 8, 8, P16, 0, 2, NULL) == -2  # C % 8 (16-bit row pieces)
    assert lib.fbbev_history_warp_vm(P16, 0, P, 1, 16, 80, 8, 8, 8, P8, 0, 2, NULL) == -2   # out not 16-byte aligned
    assert lib.fbbev_history_warp_vm(P16, 100, P, 1, 16, 80, 8, 8, 8, P16, 0, 2, NULL) == -1    # batch stride < T*N*C
    assert lib.fbbev_history_warp_vm(P16, 0, P, 1, 0, 80, 8, 8, 8, P16, 0, 2, NULL) == 0    # no frames: no-op
    assert lib.fbbev_history_frame_vm(P, 1, 80, 512, 3, P16, 0, 2, NULL) == -1              # N % inner
    assert lib.fbbev_history_frame_vm(P, 1, 82, 512, 1, P16, 0, 2, NULL) == -2              # C % 8
    assert lib.fbbev_history_conv_bf16(P16, 0, P16, P16, P16, P16, 1, 17, 32, 32, 64, P16, P16, 1 << 20, 0, 1, NULL) == -2   # C in {16, 80}
    assert lib.fbbev_history_conv_bf16(P16, 0, P16, P16, P16, P16, 1, 17, 80, 80, 64, P16, P16, 16, 0, 1, NULL) == -3        # workspace
    assert lib.fbbev_history_conv_bf16(P16, 0, P16, P16, P16, P16, 1, 17, 80, 80, 64, P16, P16, 1 << 20, 2, 1, NULL) == -1   # layout flag
    assert lib.fbbev_history_conv_bf16(P8, 0, P16, P16, P16, P16, 1, 17, 80, 80, 64, P16, P16, 1 << 20, 1, 1, NULL) == -2    # rows not aligned
    assert lib.fbbev_history_conv_vm(P16, 0, P16, P16, P16, P16, 1, 17, 80, 80, 64, P16, NULL, 0, 1, NULL) == -3             # workspace required
    assert lib.fbbev_history_conv_vm(P16, 0, P16, P16, P16, P16, 1, 17, 48, 48, 64, P16, P16, 1 << 20, 1, NULL) == -2
    assert lib.fbbev_history_conv_e(P16, 0, P16, P8, P16, P16, 1, 17, 80, 80, 64, P16, P16, 1 << 20, 1, NULL) == -2          # bias rows are 16-byte loads
    assert lib.fbbev_rank_workspace_bytes(0) == 256 and lib.fbbev_pool_dense_workspace_bytes(0, 1, 1, 1) == 256
