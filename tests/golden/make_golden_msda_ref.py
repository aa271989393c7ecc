#!/usr/bin/env python3
"""Golden vectors of the bilinear sampling / gradient functions for the MSDA oracle, produced by the REFERENCE TREE's own
code: mmdet3d/ops/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-147 (the in-tree twin of mmcv's ms_deform_attn_im2col_bilinear
/ _col2im_bilinear), compiled from where it lies by `make -C oracle ref_msda`.  Run in the container that has
/root/reference; the GPU box only sees tests/golden/msda_bilinear_ref.npz.

    python tests/golden/make_golden_msda_ref.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_msda as R  # noqa: E402


def cases(rng, n):
    out = []
    for i in range(n):
        H, W = int(rng.integers(1, 9)), int(rng.integers(1, 12))
        heads, ch = int(rng.integers(1, 5)), int(rng.integers(1, 7))
        # pixel coordinates inside the range the callers admit (-1 < x < size), with a share pinned on the borders and on
        # exact integers (floor / weight edge cases)
        h = rng.uniform(-1, H)
        w = rng.uniform(-1, W)
        k = i % 7
        if k == 1: h = float(rng.integers(0, H))
        if k == 2: w = float(rng.integers(0, W))
        if k == 3: h, w = -0.999, W - 0.001
        if k == 4: h, w = H - 0.25, -0.5
        if k == 5: h, w = float(H - 1), float(W - 1)
        # the callers hand over NORMALISED locations; pixel coordinates are loc*size - 0.5 in fp32 (two roundings): draw the
        # location and derive h, w the way mmcv's kernel (and the oracle) does, so both sides see identical coordinates
        loc = np.array([(w + 0.5) / W, (h + 0.5) / H], np.float32)
        h32 = np.float32(loc[1] * np.float32(H)) - np.float32(0.5)
        w32 = np.float32(loc[0] * np.float32(W)) - np.float32(0.5)
        if not (h32 > -1 and w32 > -1 and h32 < H and w32 < W):
            continue
        out.append((H, W, heads, ch, np.float32(h32), np.float32(w32), int(rng.integers(0, heads)), int(rng.integers(0, ch)),
                    np.float32(rng.normal()), np.float32(rng.uniform(0, 1)), loc))
    return out


def main():
    assert R.available(), 'make -C oracle ref_msda first (needs /root/reference)'
    rng = np.random.default_rng(20260924)
    rec = {k: [] for k in ('shape', 'hw', 'loc', 'mc', 'top_mask', 'data', 'sample', 'grad_im', 'grad_w', 'grad_h', 'grad_mask')}
    for H, W, heads, ch, h, w, m, c, top, mask, loc in cases(rng, 420):
        data = rng.normal(size=(H * W * heads * ch)).astype(np.float32)
        s = R.im2col(data, H, W, heads, ch, h, w, m, c)
        gi = np.zeros_like(data)
        gx, _, gm = R.col2im(data, H, W, heads, ch, h, w, m, c, float(W), top, mask, gi)      # x: offset_scale = width
        gi2 = np.zeros_like(data)
        _, gy, _ = R.col2im(data, H, W, heads, ch, h, w, m, c, float(H), top, mask, gi2)      # y: offset_scale = height
        assert np.array_equal(gi, gi2)
        rec['shape'].append([H, W, heads, ch]); rec['hw'].append([h, w]); rec['loc'].append(loc); rec['mc'].append([m, c])
        rec['top_mask'].append([top, mask]); rec['data'].append(data); rec['sample'].append(s)
        rec['grad_im'].append(gi); rec['grad_w'].append(gx); rec['grad_h'].append(gy); rec['grad_mask'].append(gm)
    np.savez_compressed(os.path.join(HERE, 'msda_bilinear_ref.npz'),
                        shape=np.array(rec['shape'], np.int32), hw=np.array(rec['hw'], np.float32), loc=np.array(rec['loc'], np.float32),
                        mc=np.array(rec['mc'], np.int32), top_mask=np.array(rec['top_mask'], np.float32),
                        data=np.concatenate(rec['data']), data_off=np.cumsum([0] + [d.size for d in rec['data']]).astype(np.int64),
                        sample=np.array(rec['sample'], np.float32), grad_im=np.concatenate(rec['grad_im']),
                        grad_w=np.array(rec['grad_w'], np.float32), grad_h=np.array(rec['grad_h'], np.float32),
                        grad_mask=np.array(rec['grad_mask'], np.float32))
    print('wrote msda_bilinear_ref.npz:', len(rec['sample']), 'cases')


if __name__ == '__main__':
    main()
