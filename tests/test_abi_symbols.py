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
