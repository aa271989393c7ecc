"""`y = x W^T + b` for the (B*Q, C) row tensors of the backward projection, with a backward built for their shape.

The backward projection applies ~10 small linear layers (80 -> 64 / 80 / 128 / 320 / 512 channels) to 160 000 BEV-query rows
(BASELINE configs[2]: 200 x 200 queries, B = 4; `spatial_cross_attention_depth.py:533-540`, `transformer.py` FFN).  In
training, autograd's weight gradient of such a layer is `mm(gy^T, x)` with K = 160 000 and an 80 x 80 ... 512 x 80 result: the
vendor GEMM runs it as 9-60 output tiles with the whole K loop inside each -- 0.3-0.45 ms per layer for a few GFLOP, 20x
off the time its 100 MB of operands take to read (profiles/r03_time_train_BL2_B4_L4_sites.json: 4 ms of the 20.6 ms step).
Here the rows are cut into S slices, the S partial products are one batched GEMM (S x tiles workgroups) and their sum a
small reduction; the bias gradient is reduced the same way in two stages.  fp32 throughout; the sums are re-associated
(slice partials), i.e. equal to autograd's within fp32 rounding, which is what the training parity tests allow."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi

MIN_ROWS = 16384            # below this autograd's plain GEMM is as good
# Inference: the split-operand bf16-MFMA kernel (fbbev_rows_linear_x3: ~1e-5 relative to fp32, bias / ReLU in its epilogue)
# instead of the vendor fp32 GEMM.  FBBEV_ROWS_LINEAR=f32 (or X3 = False) keeps the vendor GEMM.
X3 = os.environ.get('FBBEV_ROWS_LINEAR', 'x3') != 'f32'
X3_MIN_ROWS = 2048
# fold `query + query_pos` into the projections' row loads (fbbev_rows_linear_x3_add); FBBEV_ROWS_LINEAR_FOLD=0: a pass of its own
FOLD_ADDEND = os.environ.get('FBBEV_ROWS_LINEAR_FOLD', '1') != '0'
SLICE_ROWS = 2048           # rows per partial product


def _slices(rows):
    s = max(1, rows // SLICE_ROWS)
    return s, rows // s            # S slices of `per` rows; the remainder (< S rows) goes through a plain GEMM


def weight_grad(gy, x):
    """gy (R, O), x (R, I) -> gy^T x (O, I), split along R."""
    R = gy.shape[0]
    S, per = _slices(R)
    main = S * per
    g3 = gy[:main].view(S, per, gy.shape[1])
    x3 = x[:main].view(S, per, x.shape[1])
    gw = torch.bmm(g3.transpose(1, 2), x3).sum(0)
    if main < R:
        gw = gw + gy[main:].t() @ x[main:]
    return gw


def bias_grad(gy):
    """column sums of gy (R, O) as a batched ones-vector GEMM + a small reduction: `gy.sum(0)` runs ATen's strided
    reduce kernel at ~190 GB/s on these shapes (0.27 ms per layer at R = 160 000)"""
    R = gy.shape[0]
    S, per = _slices(R)
    main = S * per
    ones = torch.ones((1, 1, per), dtype=gy.dtype, device=gy.device).expand(S, 1, per)
    gb = torch.bmm(ones, gy[:main].view(S, per, gy.shape[1])).sum((0, 1))
    if main < R:
        gb = gb + gy[main:].sum(0)
    return gb


class _RowsLinear(torch.autograd.Function):
    """x (R, I) 2-D -> (R, O).  2-D on purpose: F.linear of a 3-D tensor returns a VIEW of its GEMM result, and autograd refuses
    in-place ops (the FFN's ReLU(inplace=True)) on a view created inside a custom Function; `linear_rows` reshapes outside."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)     # autocast-safe: fp32 in, fp32 GEMM,
    def forward(ctx, x, w, b):                                              # autocast off inside forward AND backward
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gy @ w
        if ctx.needs_input_grad[1]:
            gw = weight_grad(gy, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = bias_grad(gy)
        return gx, gw, gb


class X3Weights:
    """Split MFMA fragments of a weight matrix, rebuilt when the SOURCE tensors change (data_ptr + _version, as the other
    folded-weight caches of the package).  `transform(w, b) -> (w', b')` derives the matrix actually applied (row permutation,
    head padding); it runs under no_grad, once per version.  One instance per call site, owned by the module (never keyed on a
    temporary: a freed temporary's address can come back with version 0)."""

    def __init__(self):
        self.key = None
        self.w = self.b = self.frag = None

    def get(self, w_src, b_src, transform=None):
        key = (w_src.data_ptr(), w_src._version, None if b_src is None else (b_src.data_ptr(), b_src._version), str(w_src.device))
        if key != self.key:
            with torch.no_grad():
                w, b = (w_src, b_src) if transform is None else transform(w_src, b_src)
                w = w.detach().float().contiguous()
                self.w, self.b = w, None if b is None else b.detach().float().contiguous()
                if self.b is not None and self.b.data_ptr() % 16 != 0:      # a view into a flat parameter buffer: the kernels read the
                    self.b = self.b.clone()                                  # bias in 16-byte pieces (round 5), a private copy is aligned
                self.frag = _capi.rows_linear_x3_fragments(w)
            self.key = key
        return self


def x3_ok(x, in_features, out_features):
    """the split-operand kernel applies: inference on a GPU, fp32 rows with unit column stride, shapes it takes"""
    return (X3 and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and in_features % 8 == 0 and
            out_features % 4 == 0 and x.shape[-1] == in_features and x.numel() // max(1, in_features) >= X3_MIN_ROWS)


def _addend_rows(addend, x):
    """(P, I) rows of a positional addend for fbbev_rows_linear_x3_add: a (bs, Q, I) tensor that is the SAME (Q, I) table for
    every sample (a stride-0 expand, how the encoder hands over `query_pos`) repeats with period Q; anything else is taken row by
    row.  None if the kernel cannot read it as it is."""
    I = x.shape[-1]
    if addend.shape[-1] != I or addend.dtype != torch.float32 or not addend.is_cuda:
        return None
    if addend.dim() == 3 and addend.shape[0] > 1 and addend.stride(0) == 0:
        a = addend[0]
    elif addend.shape == x.shape and addend.is_contiguous():
        a = addend.reshape(-1, I)
    else:
        return None
    rows = x.numel() // I
    if a.stride(1) != 1 or a.stride(0) % 4 != 0 or a.data_ptr() % 16 != 0 or rows % a.shape[0] != 0:
        return None
    return a


def ln_fusable(norm, residual, x, out_features):
    """fbbev_rows_linear_x3_ln applies: `norm` is a one-dimensional affine LayerNorm over the out_features outputs (fp32, on the GPU),
    the residual (if any) has the output's shape with dense rows, out_features <= 128."""
    w, b = getattr(norm, 'weight', None), getattr(norm, 'bias', None)
    if (w is None or b is None or tuple(getattr(norm, 'normalized_shape', ())) != (out_features,) or out_features > 128 or
            w.dtype != torch.float32 or not w.is_cuda or w.data_ptr() % 16 != 0 or b.data_ptr() % 16 != 0):
        return False           # (LayerNorm parameters that are views into a flat buffer may be misaligned -> the two-kernel path)
    if residual is not None and (residual.shape[:-1] != x.shape[:-1] or residual.shape[-1] != out_features or
                                 residual.dtype != torch.float32 or not residual.is_contiguous() or residual.data_ptr() % 16 != 0):
        return False           # (a storage-offset residual takes the separate LayerNorm: ADVICE r4)
    return True


def linear_x3(x, cache, relu=False, out=None, addend=None, ln=None):
    """x (..., I) [+ addend] -> (..., O) through fbbev_rows_linear_x3 with the fragments of `cache` (an X3Weights after .get()).
    ln = (residual or None, LayerNorm module): LayerNorm(x W^T + b + residual) in the same kernel (fbbev_rows_linear_x3_ln)."""
    I = x.shape[-1]
    O = cache.w.shape[0]
    if ln is not None:
        assert not relu and out is None and addend is None
        res, norm = ln
        x2 = x.reshape(-1, I)
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        y = _capi.rows_linear_x3_ln(x2, cache.frag, cache.b, O, None if res is None else res.reshape(-1, O), norm.weight, norm.bias,
                                    norm.eps)
        return y.view(*x.shape[:-1], O)
    a = None
    if addend is not None:
        a = _addend_rows(addend, x) if FOLD_ADDEND else None
        if a is None:
            x = x + addend
    x2 = x.reshape(-1, I)
    if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    y = _capi.rows_linear_x3(x2, cache.frag, cache.b, O, relu=relu, out=out, addend=a)
    return y if out is not None else y.view(*x.shape[:-1], O)


def linear_rows(x, w, b=None, cache=None, transform=None, relu=False, addend=None, ln=None):
    """F.linear (+ ReLU) of x [+ addend] for row tensors.  Inference on a GPU with a `cache` (X3Weights owned by the calling module;
    `w` / `b` are then the SOURCE parameters and `transform` derives the applied matrix): the split-operand MFMA kernel, which
    also folds the addend (query_pos) into its row loads.  Training on a GPU: the split-K backward when it pays.  Otherwise F.linear."""
    if cache is not None:
        O = w.shape[0] if transform is None else None
        if x3_ok(x, x.shape[-1], O if O is not None else 4):
            c = cache.get(w, b, transform)
            if c.w.shape[1] == x.shape[-1] and c.w.shape[0] % 4 == 0:
                if ln is not None and ln_fusable(ln[1], ln[0], x, c.w.shape[0]) and not relu and addend is None:
                    return linear_x3(x, c, ln=ln)
                y = linear_x3(x, c, relu=relu, addend=addend)
                return y if ln is None else ln[1](y, ln[0])
    if addend is not None:
        x = x + addend
    if transform is not None:
        w, b = transform(w, b)
    rows = x.numel() // max(1, x.shape[-1])
    if (x.is_cuda and rows >= MIN_ROWS and torch.is_grad_enabled() and x.dtype == torch.float32 and
            (w.requires_grad or x.requires_grad or (b is not None and b.requires_grad))):
        y = _RowsLinear.apply(x.reshape(rows, x.shape[-1]), w, b).view(*x.shape[:-1], w.shape[0])
    else:
        y = F.linear(x, w, b)
    y = torch.relu_(y) if relu else y
    return y if ln is None else ln[1](y, ln[0])


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict) whose forward goes through `linear_rows`."""

    def forward(self, x, relu=False, addend=None, ln=None):
        """ln = (residual or None, LayerNorm): returns norm(linear(x) + residual) -- one kernel on the inference route"""
        if not hasattr(self, '_x3'):
            self._x3 = X3Weights()
        return linear_rows(x, self.weight, self.bias, cache=self._x3, relu=relu, addend=addend, ln=ln)
