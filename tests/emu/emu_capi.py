"""Test-only adapter: drive the C ABI of the CPU-emulated kernel library with CPU tensors.

The emulated library is the SAME capi.hip + kernel headers as the product, compiled against
tests/emu/rt.h.  This module lives under tests/ and is not importable from fb_bev_amd.
"""
import ctypes
import os
import sys
from ctypes import c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402
from fb_bev_amd import _capi  # noqa: E402  (signature table only)

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _capi.declare(ctypes.CDLL(build_emu.build()))
    return _lib


def p(t):
    assert t.is_contiguous() and not t.is_cuda
    return c_void_p(t.data_ptr())


def ok(code):
    assert code == 0, f'C ABI returned {code}'


def rank_build(coor, lower3, interval3, grid_size3, with_rank=True, depth=None, depth_threshold=0.01):
    B, N, D, H, W, _ = coor.shape
    n = B * N * D * H * W
    rb, rd, rf = (torch.full((n,), -7, dtype=torch.int32) for _ in range(3))
    st, ln, ir = (torch.full((n,), -7, dtype=torch.int32) for _ in range(3))
    counts = torch.full((2,), -1, dtype=torch.int32)
    ws = torch.zeros(lib().fbbev_rank_workspace_bytes(n), dtype=torch.uint8)
    arr = ctypes.c_float * 3
    lo, it, gs = arr(*lower3), arr(*interval3), arr(*grid_size3)
    tail = (B, N, D, H, W, ctypes.cast(lo, c_void_p), ctypes.cast(it, c_void_p), ctypes.cast(gs, c_void_p), p(rb), p(rd),
            p(rf), p(st), p(ln), p(ir) if with_rank else c_void_p(0), p(counts), p(ws), ws.numel(), None)
    if depth is None:
        ok(lib().fbbev_rank_build(p(coor), *tail))
    else:
        ok(lib().fbbev_rank_build_depth(p(coor), p(depth), depth_threshold, *tail))
    return rb, rd, rf, st, ln, ir, counts


def pool_fwd(depth, feat, out, rd, rf, rb, st, ln, n=None):
    ok(lib().fbbev_bev_pool_v2_fwd(feat.shape[-1], st.numel() if n is None else n, p(depth), p(feat), p(rd),
                                   p(rf), p(rb), p(st), p(ln), p(out), None))


def pool_bwd(out_grad, depth_grad, feat_grad, depth, feat, rd, rf, rb, st, ln):
    ok(lib().fbbev_bev_pool_v2_bwd(out_grad.shape[-1], st.numel(), p(out_grad), p(depth), p(feat), p(rd),
                                   p(rf), p(rb), p(st), p(ln), p(depth_grad), p(feat_grad), None))


def pool_zmean(depth, feat, rd, rf, ir, st, ln, counts, n_max, B, C, Z, Y, X, tile_voxels, flags=0, z_groups=1):
    out = torch.full((B, C, Y, X), float('nan'))
    ws = torch.zeros(lib().fbbev_pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8)
    ok(lib().fbbev_pool_tile_index(p(ir), p(st), p(counts), n_max, B, Z, Y, X, tile_voxels, flags, p(ws), ws.numel(), None))
    if z_groups > 1:
        partial = torch.full((z_groups * out.numel() + 4,), float('nan'))
        off = (-partial.data_ptr() // 4) % 4
        code = lib().fbbev_pool_zmean_split(p(depth), p(feat), p(rd), p(rf), p(ir), p(st), p(ln), B, C, Z, Y, X, p(out), p(ws),
                                            ws.numel(), tile_voxels, flags, z_groups, c_void_p(partial.data_ptr() + 4 * off),
                                            z_groups * out.numel() * 4, None)
        return code, out
    code = lib().fbbev_pool_zmean(p(depth), p(feat), p(rd), p(rf), p(ir), p(st), p(ln), B, C, Z, Y, X, p(out), p(ws),
                                  ws.numel(), tile_voxels, flags, None)
    return code, out


def pool_zmean_rows(depth, feat, rd, rf, ir, st, ln, counts, n_max, B, C, Z, Y, X, tile_voxels, flags=0, row_bias=None):
    out = torch.full((B, Y * X, C), float('nan'))
    ws = torch.zeros(lib().fbbev_pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8)
    ok(lib().fbbev_pool_tile_index(p(ir), p(st), p(counts), n_max, B, Z, Y, X, tile_voxels, flags, p(ws), ws.numel(), None))
    code = lib().fbbev_pool_zmean_rows(p(depth), p(feat), p(rd), p(rf), p(ir), p(st), p(ln), B, C, Z, Y, X,
                                       p(row_bias) if row_bias is not None else None, p(out), p(ws), ws.numel(), tile_voxels, flags, None)
    return code, out


def pool_dense(depth, feat, rd, rf, ir, st, ln, counts, n_max, B, C, Z, Y, X, tile_voxels, flags=0, addend=None):
    cl = bool(flags & 0x100000)
    dt = torch.bfloat16 if flags & 0x800000 else (torch.float16 if flags & 0x1000000 else torch.float32)
    out = torch.full((B, Z, Y, X, C) if cl else (B, C, Z, Y, X), float('nan'), dtype=dt)
    ws = torch.zeros(lib().fbbev_pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8)
    ok(lib().fbbev_pool_tile_index(p(ir), p(st), p(counts), n_max, B, Z, Y, X, tile_voxels, flags, p(ws), ws.numel(), None))
    if addend is not None:
        code = lib().fbbev_bev_pool_v2_dense_fwd_add(p(depth), p(feat), p(rd), p(rf), p(ir), p(st), p(ln), B, C, Z, Y, X,
                                                     p(out), 0, 0, p(ws), ws.numel(), tile_voxels, flags, p(addend), None)
        return code, out
    code = lib().fbbev_bev_pool_v2_dense_fwd(p(depth), p(feat), p(rd), p(rf), p(ir), p(st), p(ln),
                                             B, C, Z, Y, X, p(out), 0, 0, p(ws), ws.numel(), tile_voxels, flags, None)
    return code, out


def pool_dense_bwd(out_grad, depth, feat, rd, ir, st, counts, n_max, grid_zyx, poison=True, zgrad=None, zscale=0.0):
    """out_grad (B,C,Z,Y,X) (may have padded batch/channel strides) -> depth_grad, feat_grad"""
    B, N, D, H, W = depth.shape
    C = feat.shape[-1]
    Z, Y, X = grid_zyx
    dg = torch.full_like(depth, float('nan'))
    fg = torch.full_like(feat, float('nan'))
    ws = torch.zeros(lib().fbbev_pool_dense_bwd_workspace_bytes(B, N, D, H, W, C, Z, Y, X), dtype=torch.uint8)
    assert out_grad.stride()[2:] == (Y * X, X, 1)
    if zgrad is not None:      # + a (B,C,Y,X) gradient every z plane receives (fbbev_bev_pool_v2_dense_bwd_z)
        code = lib().fbbev_bev_pool_v2_dense_bwd_z(c_void_p(out_grad.data_ptr()), out_grad.stride(0), out_grad.stride(1), p(zgrad),
                                                   float(zscale), p(depth), p(feat), p(rd), p(ir), p(st), p(counts), n_max,
                                                   B, N, D, H, W, C, Z, Y, X, p(dg), p(fg), p(ws), ws.numel(), None)
        return code, dg, fg
    code = lib().fbbev_bev_pool_v2_dense_bwd(c_void_p(out_grad.data_ptr()), out_grad.stride(0), out_grad.stride(1),
                                             p(depth), p(feat), p(rd), p(ir), p(st), p(counts), n_max,
                                             B, N, D, H, W, C, Z, Y, X, p(dg), p(fg), p(ws), ws.numel(), None)
    return code, dg, fg


def msda_fwd(value, ss, ls, loc, w):
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    out = torch.full((B, Q, M * Dh), float('nan'))
    ok(lib().fbbev_msda_fwd(p(value), p(ss), p(ls), p(loc), p(w), B, S, M, Dh, L, Q, P, p(out), None))
    return out


def msda_bwd(value, ss, ls, loc, w, go):
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    gv, gl, gw = torch.zeros_like(value), torch.zeros_like(loc), torch.zeros_like(w)
    ok(lib().fbbev_msda_bwd(p(value), p(ss), p(ls), p(loc), p(w), p(go), B, S, M, Dh, L, Q, P, p(gv), p(gl),
                            p(gw), None))
    return gv, gl, gw


def msda_bwd_ws(value, ss, ls, loc, w, go, level_hw):
    """the band-binned fixed-point backward; grad_value starts as NaN (the kernels must write every token); returns None when
    the library reports no plan for the shape"""
    B, S, M, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    flat = [int(x) for hw in level_hw for x in hw]
    arr = (ctypes.c_int32 * len(flat))(*flat)
    need = lib().fbbev_msda_bwd_ws_bytes(B, S, M, Dh, L, Q, P, arr)
    if not need:
        return None
    ws = torch.zeros(need, dtype=torch.uint8)
    gv, gl, gw = torch.full_like(value, float('nan')), torch.zeros_like(loc), torch.zeros_like(w)
    ok(lib().fbbev_msda_bwd_ws(p(value), p(ss), p(ls), p(loc), p(w), p(go), B, S, M, Dh, L, Q, P, p(gv), p(gl), p(gw), arr,
                               p(ws), need, None))
    return gv, gl, gw


def lidar_coor(xs, ys, ds, cam):
    rots, trans, intrins, post_rots, post_trans, bda = cam
    B, N = trans.shape[:2]
    coor = torch.full((B, N, ds.numel(), ys.numel(), xs.numel(), 3), float('nan'))
    ok(lib().fbbev_lidar_coor(p(xs), p(ys), p(ds), p(rots), p(trans), p(intrins), p(post_rots), p(post_trans),
                              p(bda), B, N, ds.numel(), ys.numel(), xs.numel(), p(coor), None))
    return coor


def da_cross_attn_fwd(value, ss, ls, pred_depth, ref_cam, mask, qdepth, offsets, attn, d0, dstep, misalign=False,
                      head_minor=0, head_dim=None, zero_token=None, bev_w=0):
    """misalign=True: output buffer at a 4-byte (not 8-byte) aligned address => the channel-per-lane kernel runs"""
    Ncam, B, Q, Za = mask.shape
    _, S, M, HS = value.shape
    Dh = HS if head_dim is None else head_dim
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    buf = torch.full((B * Q * M * Dh + 3,), float('nan'))
    off = 1 if (buf.data_ptr() % 8 == 0) == misalign else 0
    if not misalign and (buf.data_ptr() + 4 * off) % 8:
        off += 1
    slots = buf[off:off + B * Q * M * Dh].view(B, Q, M * Dh)
    assert (slots.data_ptr() % 8 != 0) == misalign
    m8 = mask.to(torch.uint8).contiguous()
    if value.dtype != torch.float32:
        et = {torch.bfloat16: 1, torch.float16: 2}[value.dtype]
        ok(lib().fbbev_da_cross_attn_fwd_e(p(value), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(offsets),
                                           p(attn), B, Ncam, S, M, Dh, L, Q, P, Za, pred_depth.shape[1], d0, dstep,
                                           int(head_minor), HS, et, c_void_p(slots.data_ptr()), None))
        return slots.clone()
    fn, extra = lib().fbbev_da_cross_attn_fwd, ()
    if zero_token is not None:
        extra = (int(bev_w),)
        # the pipelined entry: value rows followed by ONE more token; `zero_token` = its fill (0.0 by contract; a test passes
        # another value to prove that padded corners / out-of-image samples really read it)
        buf_v = torch.empty(value.numel() + M * HS)
        buf_v[:value.numel()] = value.reshape(-1)
        buf_v[value.numel():] = zero_token
        value = buf_v[:value.numel()].view(value.shape)
        fn = lib().fbbev_da_cross_attn_fwd_zt
    ok(fn(p(value), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(offsets),
          p(attn), B, Ncam, S, M, Dh, L, Q, P, Za, pred_depth.shape[1], d0, dstep,
          int(head_minor), HS, *extra, c_void_p(slots.data_ptr()), None))
    return slots.clone()


def tokens_from_nchw_levels(levels, bias=None):
    n, C = levels[0].shape[:2]
    hws = [int(t.shape[2]) for t in levels]
    out = torch.full((n, sum(hws), C), float('nan'))
    ptrs = (c_void_p * len(levels))(*[t.data_ptr() for t in levels])
    hw = (ctypes.c_int32 * len(levels))(*hws)
    code = lib().fbbev_tokens_from_nchw_levels(ptrs, hw, len(levels), p(out), n, C, p(bias) if bias is not None else None,
                                               bias.shape[0] if bias is not None else 0, None)
    return code, out


def point_sampling(xs, ys, zs, cam, ogfH, ogfW):
    rots, trans, intrins, post_rots, post_trans, bda = cam
    B, N = trans.shape[:2]
    Q, Za = ys.numel() * xs.numel(), zs.numel()
    ref_cam = torch.full((N, B, Q, Za, 2), float('nan'))
    mask = torch.full((N, B, Q, Za), 7, dtype=torch.uint8)
    qd = torch.full((N, B, Q, Za), float('nan'))
    ok(lib().fbbev_point_sampling(p(xs), p(ys), p(zs), p(rots), p(trans), p(intrins), p(post_rots), p(post_trans),
                                  p(bda), B, N, ys.numel(), xs.numel(), Za, float(ogfH), float(ogfW), p(ref_cam), p(mask),
                                  p(qd), None))
    return ref_cam, mask.bool(), qd


def lift_rank_build(xs, ys, ds, cam, lower3, interval3, grid_size3, frustum=None, cache=None):
    """cache: dict kept by the caller across calls -> the camera-keyed entry point (same buffers every call)."""
    rots, trans, intrins, post_rots, post_trans, bda = cam
    B, N = trans.shape[:2]
    D, H, W = ds.numel(), ys.numel(), xs.numel()
    n = B * N * D * H * W
    if cache is not None and 'bufs' in cache:
        rb, rd, rf, st, ln, ir, counts, ws = cache['bufs']
    else:
        rb, rd, rf = (torch.full((n,), -7, dtype=torch.int32) for _ in range(3))
        st, ln, ir = (torch.full((n,), -7, dtype=torch.int32) for _ in range(3))
        counts = torch.full((2,), -1, dtype=torch.int32)
        ws = torch.full((lib().fbbev_rank_workspace_bytes(n),), 0xA5, dtype=torch.uint8)   # garbage: the build clears its own state
    arr = ctypes.c_float * 3
    lo, it, gs = arr(*lower3), arr(*interval3), arr(*grid_size3)
    args = [p(frustum) if frustum is not None else c_void_p(0), p(xs), p(ys), p(ds), p(rots), p(trans), p(intrins), p(post_rots), p(post_trans), p(bda),
            B, N, D, H, W, ctypes.cast(lo, c_void_p), ctypes.cast(it, c_void_p),
            ctypes.cast(gs, c_void_p), p(rb), p(rd), p(rf), p(st), p(ln), p(ir), p(counts), p(ws), ws.numel()]
    if cache is None:
        ok(lib().fbbev_lift_rank_build(*args, None))
    else:
        if 'bufs' not in cache:
            cache['bufs'] = (rb, rd, rf, st, ln, ir, counts, ws)
            cache['key'] = torch.full((lib().fbbev_cam_key_words(B, N),), -1, dtype=torch.int32)
            cache['state'] = torch.zeros(2, dtype=torch.int32)
        ok(lib().fbbev_lift_rank_build_cached(*args, p(cache['key']), p(cache['state']), None))
    return rb, rd, rf, st, ln, ir, counts


def nchw_to_nhwc(x):
    B, N, C, H, W = x.shape
    out = torch.full((B, N, H, W, C), float('nan'))
    ok(lib().fbbev_nchw_to_nhwc(p(x), p(out), B * N, C, H * W, None))
    return out


def history_flow(hist_augs, ego, bda, dx3, lower3):
    B = bda.shape[0]
    flow = torch.full((B, 4, 4), float('nan'))
    arr = ctypes.c_float * 3
    d, lo = arr(*[float(v) for v in dx3]), arr(*[float(v) for v in lower3])
    ok(lib().fbbev_history_flow(p(hist_augs), p(ego), p(bda), ctypes.cast(d, c_void_p), ctypes.cast(lo, c_void_p), B,
                                p(flow), None))
    return flow


def history_warp(history, flow, out=None):
    B, CH, Z, Y, X = history.shape
    if out is None:
        out = torch.full((B, CH, Z, Y, X), float('nan'), dtype=history.dtype)
    et = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[history.dtype]
    assert out.dtype == history.dtype
    ok(lib().fbbev_history_warp_e(c_void_p(history.data_ptr()), history.stride(0), p(flow), B, CH, Z, Y, X,
                                  c_void_p(out.data_ptr()), out.stride(0), et, None))
    return out


def layernorm(x, weight, bias, eps, residual=None):
    C = x.shape[-1]
    out = torch.full_like(x, float('nan'))
    ok(lib().fbbev_layernorm(p(x), p(residual) if residual is not None else None, p(weight), p(bias), eps,
                             x.numel() // C, C, p(out), None))
    return out


def rows_linear_x3(x, weight, bias, relu=False, out=None, addend=None):
    O, I = weight.shape
    need = lib().fbbev_rows_linear_x3_fragment_bytes(I, O)
    frag = torch.zeros(need + 16, dtype=torch.uint8)
    off = (-frag.data_ptr()) % 16
    fp = c_void_p(frag.data_ptr() + off)
    ok(lib().fbbev_rows_linear_x3_fragments(p(weight), I, O, fp, need, None))
    R = x.shape[0]
    if out is None:
        out = torch.full((R, O), float('nan'))
    b = p(bias) if bias is not None else None
    if addend is None:
        code = lib().fbbev_rows_linear_x3(c_void_p(x.data_ptr()), x.stride(0), fp, b, R, I, O, 1 if relu else 0,
                                          c_void_p(out.data_ptr()), out.stride(0), None)
    else:
        code = lib().fbbev_rows_linear_x3_add(c_void_p(x.data_ptr()), x.stride(0), c_void_p(addend.data_ptr()), addend.stride(0),
                                              addend.shape[0], fp, b, R, I, O, 1 if relu else 0, c_void_p(out.data_ptr()),
                                              out.stride(0), None)
    return code, out


def _fragments(weight):
    O, I = weight.shape
    need = lib().fbbev_rows_linear_x3_fragment_bytes(I, O)
    frag = torch.zeros(need + 16, dtype=torch.uint8)
    off = (-frag.data_ptr()) % 16
    fp = c_void_p(frag.data_ptr() + off)
    ok(lib().fbbev_rows_linear_x3_fragments(p(weight), I, O, fp, need, None))
    return frag, fp


def rows_linear_x3_ln(x, weight, bias, residual, ln_w, ln_b, eps):
    frag, fp = _fragments(weight)
    R, I = x.shape
    O = weight.shape[0]
    out = torch.full((R, O), float('nan'))
    code = lib().fbbev_rows_linear_x3_ln(c_void_p(x.data_ptr()), x.stride(0), fp, p(bias) if bias is not None else None, R, I, O,
                                         c_void_p(residual.data_ptr()) if residual is not None else None,
                                         residual.stride(0) if residual is not None else 0, p(ln_w), p(ln_b), eps,
                                         c_void_p(out.data_ptr()), out.stride(0), None)
    return code, out


def rows_ffn_x3(x, w1, b1, w2, b2, residual=None, ln_w=None, ln_b=None, eps=1e-5):
    f1, p1 = _fragments(w1)
    f2, p2 = _fragments(w2)
    R, I = x.shape
    H, O = w1.shape[0], w2.shape[0]
    out = torch.full((R, O), float('nan'))
    code = lib().fbbev_rows_ffn_x3(c_void_p(x.data_ptr()), x.stride(0), p1, p(b1), p2, p(b2), R, I, H, O,
                                   c_void_p(residual.data_ptr()) if residual is not None else None,
                                   residual.stride(0) if residual is not None else 0, p(ln_w) if ln_w is not None else None,
                                   p(ln_b) if ln_b is not None else None, eps, p(out), out.stride(0), None)
    return code, out


def rows_tail_ffn_x3(x, w0, b0, res0, ln0_w, ln0_b, eps0, w1, b1, w2, b2, ln1_w, ln1_b, eps1, tokens_per_image=None):
    f0, p0 = _fragments(w0)
    f1, p1 = _fragments(w1)
    f2, p2 = _fragments(w2)
    R, E = x.shape
    H = w1.shape[0]
    if tokens_per_image:
        out = torch.full((R // tokens_per_image, E, tokens_per_image), float('nan'))
        code = lib().fbbev_rows_tail_ffn_x3_planes(c_void_p(x.data_ptr()), x.stride(0), p0, p(b0),
                                                   c_void_p(res0.data_ptr()) if res0 is not None else None,
                                                   res0.stride(0) if res0 is not None else 0, p(ln0_w), p(ln0_b), eps0, p1, p(b1), p2, p(b2),
                                                   R, E, H, p(ln1_w), p(ln1_b), eps1, tokens_per_image, p(out), None)
        return code, out
    out = torch.full((R, E), float('nan'))
    code = lib().fbbev_rows_tail_ffn_x3(c_void_p(x.data_ptr()), x.stride(0), p0, p(b0),
                                        c_void_p(res0.data_ptr()) if res0 is not None else None, res0.stride(0) if res0 is not None else 0,
                                        p(ln0_w), p(ln0_b), eps0, p1, p(b1), p2, p(b2), R, E, H, p(ln1_w), p(ln1_b), eps1, p(out),
                                        out.stride(0), None)
    return code, out


def rows_linear_x3_planes(x, weight, bias, tokens_per_image, heads, head_dim, dtype=None):
    frag, fp = _fragments(weight)
    R, I = x.shape
    if dtype is not None:                  # 16-bit planes: fbbev_rows_linear_x3_planes_e
        out = torch.zeros((R // tokens_per_image, heads, tokens_per_image, head_dim), dtype=dtype)
        code = lib().fbbev_rows_linear_x3_planes_e(c_void_p(x.data_ptr()), x.stride(0), fp, p(bias) if bias is not None else None, R, I,
                                                   heads * head_dim, tokens_per_image, head_dim, 1 if dtype == torch.bfloat16 else 2,
                                                   p(out), None)
        return code, out
    out = torch.full((R // tokens_per_image, heads, tokens_per_image, head_dim), float('nan'))
    code = lib().fbbev_rows_linear_x3_planes(c_void_p(x.data_ptr()), x.stride(0), fp, p(bias) if bias is not None else None, R, I,
                                             heads * head_dim, tokens_per_image, head_dim, p(out), None)
    return code, out


def rows_to_head_planes(rows, tokens_per_image, heads, head_dim):
    R = rows.shape[0]
    out = torch.full((R // tokens_per_image, heads, tokens_per_image, head_dim), float('nan'))
    ok(lib().fbbev_rows_to_head_planes(p(rows), R, tokens_per_image, heads, head_dim, p(out), None))
    return out


def da_cross_attn_fused(planes, ss, ls, pred_depth, ref_cam, mask, qdepth, query, addend, w_so, b_so, w_aw, b_aw, P, d0, dstep, bev_w,
                        min_level_width=None, out_proj=None):
    Ncam, B, Q, Za = mask.shape
    BN, M, S, Dh = planes.shape
    L = ss.shape[0]
    f_so, p_so = _fragments(w_so)
    f_aw, p_aw = _fragments(w_aw)
    slots = torch.full((B, Q, M * Dh), float('nan'))
    m8 = mask.to(torch.uint8).contiguous()
    if min_level_width is None:
        min_level_width = int(ss[:, 1].min())
    a = (c_void_p(addend.data_ptr()), addend.stride(0), addend.shape[0]) if addend is not None else (None, 0, 1)
    if out_proj is not None:      # (weight, bias, residual or None, ln_weight, ln_bias, eps): fbbev_da_cross_attn_fused_ln
        wo, bo, res, lnw, lnb, eps = out_proj
        f_o, p_o = _fragments(wo)
        code = lib().fbbev_da_cross_attn_fused_ln(p(planes), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(query),
                                                  query.stride(1), *a, p_so, p(b_so), p_aw, p(b_aw), p_o, p(bo),
                                                  None if res is None else p(res), M * Dh, p(lnw), p(lnb), eps, B, Ncam, S, M, Dh, L, Q, P,
                                                  Za, pred_depth.shape[1], d0, dstep, bev_w, min_level_width, p(slots), None)
        return code, slots
    if planes.dtype in (torch.bfloat16, torch.float16):      # fbbev_da_cross_attn_fused_e: 16-bit head planes
        code = lib().fbbev_da_cross_attn_fused_e(p(planes), 1 if planes.dtype == torch.bfloat16 else 2, p(ss), p(ls), p(pred_depth), p(ref_cam),
                                                 p(m8), p(qdepth), p(query), query.stride(1), *a, p_so, p(b_so), p_aw, p(b_aw), B, Ncam, S, M,
                                                 Dh, L, Q, P, Za, pred_depth.shape[1], d0, dstep, bev_w, min_level_width, p(slots), None)
        return code, slots
    code = lib().fbbev_da_cross_attn_fused(p(planes), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(query),
                                           query.stride(1), *a, p_so, p(b_so), p_aw, p(b_aw), B, Ncam, S, M, Dh, L, Q, P, Za,
                                           pred_depth.shape[1], d0, dstep, bev_w, min_level_width, p(slots), None)
    return code, slots


def msda_self_fused(planes, ref, query, addend, w_so, b_so, w_aw, b_aw, P, bev_w, level_hw, out_proj=None):
    """out_proj = (weight, bias, residual or None, ln_weight, ln_bias, eps): fbbev_msda_self_fused_ln"""
    B, M, S, Dh = planes.shape
    Q = query.shape[1]
    f_so, p_so = _fragments(w_so)
    f_aw, p_aw = _fragments(w_aw)
    out = torch.full((B, Q, M * Dh), float('nan'))
    a = (c_void_p(addend.data_ptr()), addend.stride(0), addend.shape[0]) if addend is not None else (None, 0, 1)
    if out_proj is not None:
        wo, bo, res, lnw, lnb, eps = out_proj
        f_o, p_o = _fragments(wo)
        code = lib().fbbev_msda_self_fused_ln(p(planes), p(ref), p(query), query.stride(1), *a, p_so, p(b_so), p_aw, p(b_aw), p_o, p(bo),
                                              None if res is None else p(res), M * Dh, p(lnw), p(lnb), eps, B, S, M, Dh, 1, Q, P, bev_w,
                                              level_hw[0], level_hw[1], p(out), None)
        return code, out
    code = lib().fbbev_msda_self_fused(p(planes), p(ref), p(query), query.stride(1), *a, p_so, p(b_so), p_aw, p(b_aw), B, S, M, Dh,
                                       1, Q, P, bev_w, level_hw[0], level_hw[1], p(out), None)
    return code, out


def layernorm_bwd(x, grad_out, weight, eps):
    C = x.shape[-1]
    rows = x.numel() // C
    n = lib().fbbev_layernorm_bwd_partials(rows)
    partial = torch.full((n, 2, C), float('nan'))
    gx = torch.full_like(x, float('nan'))
    ok(lib().fbbev_layernorm_bwd(p(x), p(grad_out), p(weight), eps, rows, C, p(gx), p(partial), None))
    s = partial.sum(0)
    return gx, s[0], s[1]


def da_cross_attn_fwd_planes(value, ss, ls, pred_depth, ref_cam, mask, qdepth, offsets, attn, d0, dstep, head_minor=0, head_dim=None,
                             bev_w=0, min_level_width=2):
    """value rows -> head planes (fbbev_value_rows_to_head_planes) -> fbbev_da_cross_attn_fwd_planes; returns (code, slots)"""
    Ncam, B, Q, Za = mask.shape
    BN, S, M, HS = value.shape
    Dh = HS if head_dim is None else head_dim
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    planes = torch.full((BN, M, S, Dh), float('nan'))
    ok(lib().fbbev_value_rows_to_head_planes(p(value), BN * S, S, M, Dh, HS, 1 if head_minor & 4 else 0, p(planes), None))
    slots = torch.full((B, Q, M * Dh), float('nan'))
    m8 = mask.to(torch.uint8).contiguous()
    code = lib().fbbev_da_cross_attn_fwd_planes(p(planes), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(offsets), p(attn),
                                                B, Ncam, S, M, Dh, L, Q, P, Za, pred_depth.shape[1], d0, dstep, int(head_minor) & 3,
                                                int(bev_w), int(min_level_width), p(slots), None)
    return code, slots, planes


def da_cross_attn_bwd(value, ss, ls, pred_depth, ref_cam, mask, qdepth, offsets, attn, d0, dstep, grad_slots, head_minor=0,
                      head_dim=None, lds_planes=False, level_hw=None, bev_w=0):
    Ncam, B, Q, Za = mask.shape
    _, S, M, HS = value.shape
    Dh = HS if head_dim is None else head_dim
    L, P = (attn.shape[2], attn.shape[3]) if head_minor & 2 else (attn.shape[3], attn.shape[4])
    gv, gd, go, ga = (torch.zeros_like(t) for t in (value, pred_depth, offsets, attn))
    m8 = mask.to(torch.uint8).contiguous()
    args = (p(value), p(ss), p(ls), p(pred_depth), p(ref_cam), p(m8), p(qdepth), p(offsets), p(attn), p(grad_slots), B, Ncam,
            S, M, Dh, L, Q, P, Za, pred_depth.shape[1], d0, dstep, int(head_minor), HS, p(gv), p(gd), p(go), p(ga))
    if lds_planes:
        arr = _capi._level_hw(level_hw, L)
        need = lib().fbbev_da_cross_attn_bwd_ws_bytes(B, Ncam, S, M, Dh, Q, HS, L, P, arr)
        assert need > 0
        ws = torch.full((need // 4,), float('nan'))
        gv.fill_(float('nan'))                       # written, not accumulated
        if bev_w:
            ok(lib().fbbev_da_cross_attn_bwd_ws_grid(*args, arr, p(ws), need, int(bev_w), None))
        else:
            ok(lib().fbbev_da_cross_attn_bwd_ws(*args, arr, p(ws), need, None))
    else:
        ok(lib().fbbev_da_cross_attn_bwd(*args, None))
    return gv, gd, go, ga


def history_conv(feats, w1, bias1, w2, bias2, bf16=False, voxel_major=False, x3=False):
    C, Cout = w1.shape[0], w2.shape[0]
    if voxel_major:
        B, T1, N, _ = feats.shape
    else:
        B, TC, N = feats.shape
        T1 = TC // C
    out = torch.full((B, Cout, N), float('nan'))
    ws = torch.zeros((1 + T1) * C * max(C, Cout, 96) + B * T1 * C)
    et = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[feats.dtype]
    args = (c_void_p(feats.data_ptr()), feats.stride(0), p(w1), p(bias1), p(w2), p(bias2), B, T1, C, Cout, N, p(out), p(ws),
            ws.numel() * 4)
    if x3:
        ok(lib().fbbev_history_conv_bf16x3(*args, et, None))
    elif bf16:
        ok(lib().fbbev_history_conv_bf16(*args, 1 if voxel_major else 0, et, None))
    elif voxel_major:
        ok(lib().fbbev_history_conv_vm(*args, et, None))
    else:
        ok(lib().fbbev_history_conv_e(*args, et, None))
    return out


def history_warp_vm(history, flow, grid_zyx, out=None):
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    if out is None:
        out = torch.full((B, T, N, C), float('nan'), dtype=history.dtype)
    et = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[history.dtype]
    ok(lib().fbbev_history_warp_vm(c_void_p(history.data_ptr()), history.stride(0), p(flow), B, T, C, Z, Y, X,
                                   c_void_p(out.data_ptr()), out.stride(0), et, None))
    return out


def history_fused_x3_vm(history, flow, nxt, grid_zyx, w1, bias1, w2, bias2):
    """fbbev_history_fused_x3_vm: writes nxt[:, 1:] and returns (code, out (B, Cout, N)); nxt[:, 0] must hold the current frame."""
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    Cout = w2.shape[0]
    out = torch.full((B, Cout, N), float('nan'))
    ws = torch.zeros((2 + T) * C * max(C, Cout, 96) + B * (T + 1) * C + 16)
    et = {torch.bfloat16: 1, torch.float16: 2}[history.dtype]
    code = lib().fbbev_history_fused_x3_vm(c_void_p(history.data_ptr()), history.stride(0), c_void_p(nxt.data_ptr()), nxt.stride(0),
                                           p(flow), p(w1), p(bias1), p(w2), p(bias2), B, T, C, Cout, Z, Y, X, p(out), p(ws),
                                           ws.numel() * 4, et, None)
    return code, out


def history_step_x3_vm(history, flow, nxt, grid_zyx, w1, bias1, w2, bias2, chunks=0):
    """fbbev_history_step_x3_vm: writes nxt[:, 1:] and returns out (B, Cout, N); nxt[:, 0] must hold the current frame."""
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    Cout = w2.shape[0]
    out = torch.full((B, Cout, N), float('nan'))
    ws = torch.zeros((2 + T) * C * max(C, Cout, 96) + B * (T + 1) * C)
    et = {torch.bfloat16: 1, torch.float16: 2}[history.dtype]
    ok(lib().fbbev_history_step_x3_vm(c_void_p(history.data_ptr()), history.stride(0), c_void_p(nxt.data_ptr()), nxt.stride(0),
                                      p(flow), p(w1), p(bias1), p(w2), p(bias2), B, T, C, Cout, Z, Y, X, p(out), p(ws),
                                      ws.numel() * 4, et, chunks, None))
    return out


def history_fused_vm(history, flow, nxt, grid_zyx, w1, bias1, w2, bias2):
    """fbbev_history_fused_vm: writes nxt[:, 1:] and returns out (B, Cout, N); nxt[:, 0] must hold the current frame."""
    B, T, N, C = history.shape
    Z, Y, X = grid_zyx
    Cout = w2.shape[0]
    out = torch.full((B, Cout, N), float('nan'))
    ws = torch.zeros((2 + T) * C * max(C, Cout, 96))
    et = {torch.bfloat16: 1, torch.float16: 2}[history.dtype]
    code = lib().fbbev_history_fused_vm(c_void_p(history.data_ptr()), history.stride(0), c_void_p(nxt.data_ptr()), nxt.stride(0),
                                        p(flow), p(w1), p(bias1), p(w2), p(bias2), B, T, C, Cout, Z, Y, X, p(out), p(ws),
                                        ws.numel() * 4, et, None)
    return code, out


def history_frame_vm(curr, dtype, out=None, inner=1):
    B, C, N = curr.shape
    if out is None:
        out = torch.full((B, N, C), float('nan'), dtype=dtype)
    et = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    ok(lib().fbbev_history_frame_vm(p(curr), B, C, N, inner, c_void_p(out.data_ptr()), out.stride(0), et, None))
    return out


def tokens_from_nchw(x, out, out_offset=0, bias=None, pos_bias=None):
    n, C, HW = x.shape
    if pos_bias is not None:
        ok(lib().fbbev_tokens_from_nchw_pos(p(x), p(out), n, C, HW, out.stride(0), out_offset, p(pos_bias), None))
        return out
    ok(lib().fbbev_tokens_from_nchw(p(x), p(out), n, C, HW, out.stride(0), out_offset, None if bias is None else p(bias),
                                    0 if bias is None else bias.shape[0], None))
    return out


def msda_fwd_fused(value, ss, ls, ref, offsets, w, head_dim=None, offsets_head_minor=False, value_interleaved=False):
    B, S, M, HS = value.shape
    Dh = HS if head_dim is None else head_dim
    _, Q, _, L, P = w.shape
    out = torch.full((B, Q, M * Dh), float('nan'))
    ok(lib().fbbev_msda_fwd_fused(p(value), p(ss), p(ls), p(ref), p(offsets), p(w), B, S, M, Dh, L, Q, P, HS,
                                  (1 if offsets_head_minor else 0) | (4 if value_interleaved else 0), p(out), None))
    return out


def conv3d_fragments(w, transposed=False):
    """Host-side weight layout of fbbev_conv3d_ndhwc (same arithmetic as fb_bev_amd.mfma_conv3d.weight_fragments)."""
    from fb_bev_amd.mfma_conv3d import weight_fragments
    return weight_fragments(w, transposed)


def conv3d_ndhwc(x, wf, bias, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None, transposed=False):
    B, Di, Hi, Wi, Cin = x.shape
    if transposed:
        Do, Ho, Wo = Di, Hi, Wi
        out = torch.full((B, 2 * Di, 2 * Hi, 2 * Wi, Cout), float('nan'))
    else:
        Do, Ho, Wo = [(n + 2 * pad - ksize) // stride + 1 for n in (Di, Hi, Wi)]
        out = torch.full((B, Do, Ho, Wo, Cout), float('nan'))
    code = lib().fbbev_conv3d_ndhwc(p(x), p(wf), p(bias), p(residual) if residual is not None else None, B, Di, Hi, Wi, Cin,
                                    Do, Ho, Wo, Cout, ksize, stride, pad, 1 if relu else 0, 1 if transposed else 0, p(out), None)
    return code, out


def blend_levels_ndhwc(level0, coarse, wsoft):
    import ctypes
    B, D, H, W, C = level0.shape
    n = len(coarse)
    ptrs = (c_void_p * max(n, 1))(*[t.data_ptr() for t in coarse])
    dims = (ctypes.c_int * max(3 * n, 1))(*[int(v) for t in coarse for v in t.shape[1:4]])
    out = torch.full(level0.shape, float('nan'))
    code = lib().fbbev_blend_levels_ndhwc(p(level0), ctypes.cast(ptrs, c_void_p), ctypes.cast(dims, c_void_p), n, p(wsoft),
                                          int(wsoft.shape[4]), B, D, H, W, C, p(out), None)
    return code, out


def conv3d_dgrad_ndhwc(dy, wft, in_dims, Cin, ksize=3, stride=1, pad=1):
    B, Do, Ho, Wo, Cout = dy.shape
    dx = torch.full((B, *in_dims, Cin), float('nan'))
    zero = torch.zeros((Cin + 15) // 16 * 16)
    code = lib().fbbev_conv3d_dgrad_ndhwc(p(dy), p(wft), p(zero), B, Do, Ho, Wo, Cout, *in_dims, Cin, ksize, stride, pad, p(dx), None)
    return code, dx


def conv3d_wgrad_ndhwc(x, dy, ksize=3, stride=1, pad=1):
    B, Di, Hi, Wi, Cin = x.shape
    _, Do, Ho, Wo, Cout = dy.shape
    dw = torch.zeros(ksize ** 3, Cout, Cin)
    code = lib().fbbev_conv3d_wgrad_ndhwc(p(x), p(dy), B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksize, stride, pad, p(dw), None)
    return code, dw


def conv2d_nhwc(x, wf, bias, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None):
    B, Hi, Wi, Cin = x.shape
    Ho, Wo = [(n + 2 * pad - ksize) // stride + 1 for n in (Hi, Wi)]
    out = torch.full((B, Ho, Wo, Cout), float('nan'))
    code = lib().fbbev_conv2d_nhwc(p(x), p(wf), p(bias), p(residual) if residual is not None else None, B, Hi, Wi, Cin, Ho, Wo, Cout,
                                   ksize, stride, pad, 1 if relu else 0, p(out), None)
    return code, out


def conv3d_ndhwc_bf16(x, wfb, bias, Cout, ksize=3, stride=1, pad=1, relu=False, residual=None, transposed=False, planar=False):
    B, Di, Hi, Wi, Cin = x.shape
    if transposed:
        Do, Ho, Wo = Di, Hi, Wi
        out = torch.full((B, 2 * Di, 2 * Hi, 2 * Wi, Cout), float('nan'))
    else:
        Do = 1 if planar else (Di + 2 * pad - ksize) // stride + 1
        Ho, Wo = [(n + 2 * pad - ksize) // stride + 1 for n in (Hi, Wi)]
        out = torch.full((B, Do, Ho, Wo, Cout), float('nan'))
    assert wfb.dtype == torch.bfloat16 and wfb.is_contiguous()
    code = lib().fbbev_conv3d_ndhwc_bf16(p(x), c_void_p(wfb.data_ptr()), p(bias), p(residual) if residual is not None else None, B,
                                         Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksize, stride, pad, 1 if relu else 0,
                                         1 if transposed else 0, 1 if planar else 0, p(out), None)
    return code, out


def conv3d_k3s1_tiled_bf16(x, wfb, bias, Cout, relu=False, residual=None):
    B, D, H, W, Cin = x.shape
    out = torch.full((B, D, H, W, Cout), float('nan'))
    code = lib().fbbev_conv3d_k3s1_tiled_bf16(p(x), c_void_p(wfb.data_ptr()), p(bias), p(residual) if residual is not None else None,
                                              B, D, H, W, Cin, Cout, 1 if relu else 0, p(out), None)
    return code, out


def rows_wgrad_x3(grad_out, x, with_bias=True, addend=None):
    """fbbev_rows_wgrad_x3 on CPU tensors (row strides taken from the views) -> (code, grad_weight, grad_bias)"""
    R, O = grad_out.shape
    I = x.shape[1]
    need = lib().fbbev_rows_wgrad_x3_ws_bytes(R, I, O)
    ws = torch.full((need // 4 + 4,), float('nan'))
    off = ((-ws.data_ptr()) % 16) // 4
    gw = torch.full((O, I), float('nan'))
    gb = torch.full((O,), float('nan')) if with_bias else None
    code = lib().fbbev_rows_wgrad_x3(c_void_p(grad_out.data_ptr()), grad_out.stride(0), c_void_p(x.data_ptr()), x.stride(0),
                                     None if addend is None else c_void_p(addend.data_ptr()), 0 if addend is None else addend.stride(0),
                                     1 if addend is None else addend.shape[0], R, I, O,
                                     p(gw), p(gb) if gb is not None else None, c_void_p(ws.data_ptr() + 4 * off), need, None)
    return code, gw, gb


def rows_linear_x3_train(x, weight, bias, relu=False, addend=None, residual=None, mask=None, out=None):
    frag, fp = _fragments(weight)
    R, I = x.shape
    O = weight.shape[0]
    if out is None:
        out = torch.full((R, O), float('nan'))
    code = lib().fbbev_rows_linear_x3_train(
        c_void_p(x.data_ptr()), x.stride(0), None if addend is None else c_void_p(addend.data_ptr()),
        0 if addend is None else addend.stride(0), 1 if addend is None else addend.shape[0], fp, p(bias) if bias is not None else None,
        R, I, O, 1 if relu else 0, None if residual is None else c_void_p(residual.data_ptr()), 0 if residual is None else residual.stride(0),
        None if mask is None else c_void_p(mask.data_ptr()), 0 if mask is None else mask.stride(0), c_void_p(out.data_ptr()), out.stride(0), None)
    return code, out


def sum_leading(x, x2=None):
    B = x.shape[0]
    out = torch.full(x.shape[1:], float('nan'))
    code = lib().fbbev_sum_leading(p(x), p(x2) if x2 is not None else None, B, x.numel() // B, p(out), None)
    return code, out


def sum_partials(part):
    n, ln = part.shape[0], part.numel() // part.shape[0]
    out = torch.full((ln,), float('nan'))
    code = lib().fbbev_sum_partials(p(part), n, ln, p(out), None)
    return code, out


def softmax_groups(x, group):
    out = torch.full_like(x, float('nan'))
    return lib().fbbev_softmax_groups(p(x), x.numel() // group, group, p(out), None), out


def softmax_groups_bwd(y, gy, group):
    out = torch.full_like(y, float('nan'))
    return lib().fbbev_softmax_groups_bwd(p(y), p(gy), y.numel() // group, group, p(out), None), out
