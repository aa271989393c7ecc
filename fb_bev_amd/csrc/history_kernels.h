// history_kernels.h -- temporal history alignment of FB-OCC.
//
// Replaces FBOCC.generate_grid + F.grid_sample of FBOCC.fuse_history
// (mmdet3d/models/fbbev/detectors/fbocc.py:169-205, 264-275): the reference materialises a (B,Y,X,Z,4,1) homogeneous
// grid, multiplies it by a batched 4x4 (rt_flow), normalises, permutes, and hands a (B,Z,Y,X,3) grid to the
// generic 5-D grid_sample -- then cats / clones the 16-frame history (410 MB per sample at 100x100x8, fp32)
// three more times.  Here
//   k_history_flow : rt_flow[b] = inv(feat2bev) . history_forward_augs[b] . curr_to_prev_ego_rt[b]
//                    . inv(forward_augs[b]) . feat2bev   (fbocc.py:184-203), one thread per sample, closed-form
//                    inverses (feat2bev is scale+translation, forward_augs is the bda block), products
//                    associated left to right like the reference expression.
//   k_history_warp : a thread owns an output voxel (x fastest => coalesced taps for near-translations), evaluates
//                    rt_flow . (x,y,z,1), repeats the reference's normalise / un-normalise arithmetic
//                    (align_corners=True), builds the 8 trilinear taps once (zero padding: weight 0) and then
//                    streams over a group of channels: 8 gathers + 8 fma + 1 store per channel.  The sampling grid
//                    is never materialised and the output can be written straight into the frame slots of the
//                    next history buffer (out_stride_b).
// Bound: HBM (history read once through L2 + written once).
#pragma once
#include "rt.h"
#include "geom_kernels.h"
#include "pool_kernels.h"      // fbbev_f32_to_f16: the integer-only binary32 -> binary16 rounding (same bits on the emulator)

// Storage element of the history ring: ET 0 = f32, 1 = bf16, 2 = f16 (BASELINE configs[4] names fp16).  All arithmetic is
// fp32: elements are widened exactly at the load and rounded ONCE (nearest-even) at the store.
__device__ __forceinline__ float fbbev_f16_bits_to_f32(unsigned int h) {
    const unsigned int sign = (h & 0x8000u) << 16;
    unsigned int e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {                                              // subnormal half: normalise
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            u = sign | ((unsigned int)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
template <int ET>
__device__ __forceinline__ float fbbev_ld_elem(const void* base, long long i) {
    if constexpr (ET == 0) return static_cast<const float*>(base)[i];
    else {
        const unsigned int h = static_cast<const unsigned short*>(base)[i];
        if constexpr (ET == 1) { const unsigned int u = h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
        else return fbbev_f16_bits_to_f32(h);
    }
}
template <int ET>
__device__ __forceinline__ void fbbev_st_elem(void* base, long long i, float v) {
    if constexpr (ET == 0) static_cast<float*>(base)[i] = v;
    else static_cast<unsigned short*>(base)[i] = (unsigned short)(fbbev_pack2<ET>(v, 0.f) & 0xffffu);
}

__device__ __forceinline__ void fbbev_mat4(const float* a, const float* b, float* o) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            o[r * 4 + c] = a[r * 4] * b[c] + a[r * 4 + 1] * b[4 + c] + a[r * 4 + 2] * b[8 + c] + a[r * 4 + 3] * b[12 + c];
}

// dx3 = voxel size, t3 = feat2bev translation (bx - dx/2 = grid lower bound), both (x,y,z)
__global__ void __launch_bounds__(64)
k_history_flow(const float* __restrict__ hist_augs, const float* __restrict__ ego, const float* __restrict__ bda,
               float dx0, float dx1, float dx2, float t0, float t1, float t2, int B, float* __restrict__ flow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float f2b[16] = {dx0, 0.f, 0.f, t0, 0.f, dx1, 0.f, t1, 0.f, 0.f, dx2, t2, 0.f, 0.f, 0.f, 1.f};
    float inv[16] = {1.f / dx0, 0.f, 0.f, -t0 / dx0, 0.f, 1.f / dx1, 0.f, -t1 / dx1, 0.f, 0.f, 1.f / dx2, -t2 / dx2,
                     0.f, 0.f, 0.f, 1.f};
    float ib[9], fi[16], a[16], c[16];
    fbbev_inv3(bda + b * 9, ib);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) fi[r * 4 + k] = ib[r * 3 + k];
        fi[r * 4 + 3] = 0.f;
    }
    fi[12] = fi[13] = fi[14] = 0.f; fi[15] = 1.f;
    fbbev_mat4(inv, hist_augs + b * 16, a);
    fbbev_mat4(a, ego + b * 16, c);
    fbbev_mat4(c, fi, a);
    fbbev_mat4(a, f2b, c);
    for (int k = 0; k < 16; ++k) flow[b * 16 + k] = c[k];
}

// work item = ((b * n_groups) + group) * n_chunks + chunk
template <int ET>
__global__ void __launch_bounds__(256)
k_history_warp(const void* __restrict__ hist, long long hist_stride_b, const float* __restrict__ flow, int CH,
               int Z, int Y, int X, int ch_per_block, int n_groups, int n_chunks, int per_xcd, int n_work,
               void* __restrict__ out, long long out_stride_b) {
    // workgroup b runs on XCD b%8: give each XCD one contiguous eighth of the (sample, channel group, chunk) space so
    // that the y-neighbour taps of adjacent chunks are re-used from that XCD's own L2
    const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= n_work) return;
    const int chunk = work % n_chunks;
    const int bg = work / n_chunks;
    const int grp = bg % n_groups, b = bg / n_groups;
    const int YX = Y * X, ZYX = Z * YX;
    const int v = chunk * 256 + threadIdx.x;
    if (v >= ZYX) return;
    const int z = v / YX, r = v - z * YX, y = r / X, x = r - y * X;
    const float* m = flow + b * 16;
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    // fbocc.py:205 rt_flow @ (x,y,z,1); :208-209 normalise; ATen grid_sampler_unnormalize (align_corners=True)
    float gx = m[0] * fx + m[1] * fy + m[2] * fz + m[3];
    float gy = m[4] * fx + m[5] * fy + m[6] * fz + m[7];
    float gz = m[8] * fx + m[9] * fy + m[10] * fz + m[11];
    gx = gx / (float)(X - 1) * 2.0f - 1.0f;
    gy = gy / (float)(Y - 1) * 2.0f - 1.0f;
    gz = gz / (float)(Z - 1) * 2.0f - 1.0f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(X - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(Y - 1);
    const float iz = ((gz + 1.f) / 2.f) * (float)(Z - 1);
    // taps; NaN / huge coordinates fail every bounds test below => all weights 0 => output 0 (zero padding)
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const float wx1 = ix - x0f, wy1 = iy - y0f, wz1 = iz - z0f;
    const float wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy, wz0 = (z0f + 1.f) - iz;
    const bool fin = (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f) && (fabsf(iz) < 1.0e9f);
    const int x0 = fin ? (int)x0f : -2, y0 = fin ? (int)y0f : -2, z0 = fin ? (int)z0f : -2;
    int off[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // order tnw,tne,tsw,tse,bnw,bne,bsw,bse: x fastest, then y, then z
        const int cx = x0 + (k & 1), cy = y0 + ((k >> 1) & 1), cz = z0 + (k >> 2);
        const bool ok = cx >= 0 && cx < X && cy >= 0 && cy < Y && cz >= 0 && cz < Z;
        const float wk = ((k & 1) ? wx1 : wx0) * (((k >> 1) & 1) ? wy1 : wy0) * ((k >> 2) ? wz1 : wz0);
        off[k] = ok ? (cz * Y + cy) * X + cx : 0;
        w[k] = ok ? wk : 0.f;
    }
    const int c0 = grp * ch_per_block;
    const int c1 = (c0 + ch_per_block < CH) ? c0 + ch_per_block : CH;
    // element offsets of this workgroup's first channel (16-bit storage: the same offsets, half the bytes)
    long long so = (long long)b * hist_stride_b + (long long)c0 * ZYX;
    long long dof = (long long)b * out_stride_b + (long long)c0 * ZYX + v;
    int c = c0;
    constexpr int U = 4;                           // channels in flight per thread: 8*U independent gathers
    for (; c + U <= c1; c += U) {
        float a[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[u][k] = fbbev_ld_elem<ET>(hist, so + (long long)u * ZYX + off[k]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s0 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s0 = fmaf(a[u][k], w[k], s0);
            fbbev_st_elem<ET>(out, dof + (long long)u * ZYX, s0);
        }
        so += U * (long long)ZYX;
        dof += U * (long long)ZYX;
    }
    for (; c < c1; ++c) {
        float s0 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s0 = fmaf(fbbev_ld_elem<ET>(hist, so + off[k]), w[k], s0);
        fbbev_st_elem<ET>(out, dof, s0);
        so += ZYX;
        dof += ZYX;
    }
}
