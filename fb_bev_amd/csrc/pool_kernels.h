// pool_kernels.h -- bev_pool_v2 (lift-splat voxel pooling) kernels for gfx950.
//
// Reference semantics: mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:18-45 (forward),
// :64-118 (backward).  The reference maps one CUDA thread to one (interval, channel) scalar
// (forward) and one thread to a whole feature-pixel interval (backward, 17 blocks on the shipped
// config).  Here:
//   * forward "rows"  : a group of C/4 lanes owns one interval; each lane carries 4 channels as a
//     float4, so a feature row is one fully coalesced 256-320 B read, and the interval is an
//     fmaf chain in the given sorted order (bit-identical arithmetic to the reference kernel).
//   * forward "dense" : a 256-thread workgroup owns a tile of TV consecutive voxels of one (b,z)
//     plane for all C channels.  Sparse per-voxel sums are staged in an LDS tile [C][TV] (the
//     per-pillar accumulation), then the whole tile -- zeros included -- is streamed to HBM once,
//     in the final (B,C,Z,Y,X) layout, as 16-byte stores forming 4*TV-byte contiguous runs per
//     channel.  This removes the reference's new_zeros + kernel write + permute().contiguous()
//     (4x the output bytes) and is the HBM-roofline kernel of the path.
//   * backward        : one wave64 per feature-pixel interval, lanes over channels; the C-long
//     dot product for depth_grad is a wave reduction, feat_grad stays an in-order fmaf chain.
#pragma once
#include "rt.h"

template <int VEC>
__device__ __forceinline__ void fbbev_ldv(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = p[j];
    }
}

template <int VEC>
__device__ __forceinline__ void fbbev_stv(float* __restrict__ p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        float4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<float4*>(p) = t;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) p[j] = v[j];
    }
}

// In-order fmaf chain over one interval for VEC channels starting at fbase (= feat + channel offset).
// Loads for 4 points are issued together so the index -> depth/feat dependent latency overlaps.
template <int VEC>
__device__ __forceinline__ void fbbev_interval_sum(int c, int s, int len,
                                                   const float* __restrict__ depth,
                                                   const float* __restrict__ fbase,
                                                   const int* __restrict__ rd,
                                                   const int* __restrict__ rf, float (&acc)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int k = 0;
    for (; k + 4 <= len; k += 4) {
        const int pd0 = rd[s + k], pd1 = rd[s + k + 1], pd2 = rd[s + k + 2], pd3 = rd[s + k + 3];
        const int pf0 = rf[s + k], pf1 = rf[s + k + 1], pf2 = rf[s + k + 2], pf3 = rf[s + k + 3];
        const float d0 = depth[pd0], d1 = depth[pd1], d2 = depth[pd2], d3 = depth[pd3];
        float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
        fbbev_ldv<VEC>(fbase + (long long)pf0 * c, f0);
        fbbev_ldv<VEC>(fbase + (long long)pf1 * c, f1);
        fbbev_ldv<VEC>(fbase + (long long)pf2 * c, f2);
        fbbev_ldv<VEC>(fbase + (long long)pf3 * c, f3);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[j] = fmaf(f0[j], d0, acc[j]);
            acc[j] = fmaf(f1[j], d1, acc[j]);
            acc[j] = fmaf(f2[j], d2, acc[j]);
            acc[j] = fmaf(f3[j], d3, acc[j]);
        }
    }
    for (; k < len; ++k) {
        const float d0 = depth[rd[s + k]];
        float f0[VEC];
        fbbev_ldv<VEC>(fbase + (long long)rf[s + k] * c, f0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(f0[j], d0, acc[j]);
    }
}

// ---------------------------------------------------------------- forward, reference layout
// out (B,Z,Y,X,C), pre-zeroed by the caller; one lane group of c/VEC lanes per interval.
template <int VEC>
__global__ void __launch_bounds__(256)
k_pool_fwd_rows(int c, int n_intervals, const float* __restrict__ depth,
                const float* __restrict__ feat, const int* __restrict__ rd,
                const int* __restrict__ rf, const int* __restrict__ rb,
                const int* __restrict__ starts, const int* __restrict__ lengths,
                float* __restrict__ out) {
    const int slots = c / VEC;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long interval = t / slots;
    if (interval >= n_intervals) return;
    const int slot = (int)(t - interval * slots);
    const int s = starts[interval], len = lengths[interval];
    float acc[VEC];
    fbbev_interval_sum<VEC>(c, s, len, depth, feat + slot * VEC, rd, rf, acc);
    fbbev_stv<VEC>(out + (long long)rb[s] * c + slot * VEC, acc);
}

// ---------------------------------------------------------------- tile index for the dense kernel
// tile t = (plane p = b*Z+z, k) covers ranks [p*YX + k*TV, p*YX + min((k+1)*TV, YX)).
// tile_istart[t] = first interval whose rank >= the tile's first rank (lower bound), t in [0,n_tiles];
// tile_istart[n_tiles] = number of intervals with rank < total voxels.
__global__ void __launch_bounds__(256)
k_tile_lower_bound(int n_tiles, int tiles_per_plane, int YX, int TV, const int* __restrict__ rb,
                   const int* __restrict__ starts, const int* __restrict__ n_intervals_dev,
                   int n_intervals_max, int* __restrict__ tile_istart) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    int n = *n_intervals_dev;
    if (n > n_intervals_max) n = n_intervals_max;
    const int plane = t / tiles_per_plane, k = t - plane * tiles_per_plane;
    const long long target = (long long)plane * YX + (long long)k * TV;  // t==n_tiles -> total voxels
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const long long r = rb[starts[mid]];
        if (r < target) lo = mid + 1; else hi = mid;
    }
    tile_istart[t] = lo;
}

// ---------------------------------------------------------------- forward, fused dense (B,C,Z,Y,X)
// LDS tile layout: row c at c*(TV+4) floats (16-byte aligned rows for ds_read_b128 in phase 2).
// Requires C % 4 == 0, YX % 4 == 0, out 16-byte aligned.
template <int TV>
__global__ void __launch_bounds__(256)
k_pool_fwd_dense(int C, int Z, int YX, int tiles_per_plane, const float* __restrict__ depth,
                 const float* __restrict__ feat, const int* __restrict__ rd,
                 const int* __restrict__ rf, const int* __restrict__ rb,
                 const int* __restrict__ starts, const int* __restrict__ lengths,
                 const int* __restrict__ tile_istart, float* __restrict__ out) {
    constexpr int LD = TV + 4;       // LDS row stride in floats
    constexpr int Q4 = TV / 4;       // float4 per row
    float* tile = fbbev_dyn_lds_f32();
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int plane = t / tiles_per_plane, k = t - plane * tiles_per_plane;
    const int b = plane / Z, z = plane - b * Z;
    const int v0 = k * TV;
    const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
    const int i0 = tile_istart[t], i1 = tile_istart[t + 1];
    const long long cstride = (long long)Z * YX;  // floats between channels
    float* __restrict__ obase = out + ((long long)b * C * Z + z) * YX + v0;
    const int n4 = C * Q4;

    if (i0 == i1) {  // empty tile: stream zeros straight from registers, no LDS round trip
        float4 zero; zero.x = zero.y = zero.z = zero.w = 0.f;
        for (int idx = tid; idx < n4; idx += 256) {
            const int c = idx / Q4, j = (idx - c * Q4) * 4;
            if (j < nv) *reinterpret_cast<float4*>(obase + c * cstride + j) = zero;
        }
        return;
    }

    for (int idx = tid; idx < C * LD; idx += 256) tile[idx] = 0.f;
    __syncthreads();

    // phase 1: per-voxel (per-pillar) sums into the LDS tile; C/4 lanes per interval
    {
        const int lpi = C >> 2;
        const int gpb = 256 / lpi;
        const int g = tid / lpi, slot = tid - g * lpi;
        if (g < gpb) {
            const int rank0 = plane * YX + v0;
            for (int i = i0 + g; i < i1; i += gpb) {
                const int s = starts[i], len = lengths[i];
                const int v = rb[s] - rank0;
                float acc[4];
                fbbev_interval_sum<4>(C, s, len, depth, feat + slot * 4, rd, rf, acc);
                if (v >= 0 && v < nv) {
                    float* dst = tile + (slot * 4) * LD + v;
                    dst[0] = acc[0]; dst[LD] = acc[1]; dst[2 * LD] = acc[2]; dst[3 * LD] = acc[3];
                }
            }
        }
    }
    __syncthreads();

    // phase 2: stream the tile out, 16 B per lane, 4*TV-byte contiguous run per channel row
    for (int idx = tid; idx < n4; idx += 256) {
        const int c = idx / Q4, j = (idx - c * Q4) * 4;
        if (j < nv) {
            const float4 val = *reinterpret_cast<const float4*>(tile + c * LD + j);
            *reinterpret_cast<float4*>(obase + c * cstride + j) = val;
        }
    }
}

// ---------------------------------------------------------------- backward
// One wave64 per interval over ranks_feat (= one feature pixel); lane handles channels
// lane, lane+64, ... (NCH = ceil(c/64)).  Per point: depth_grad = <out_grad row, feat row>
// (wave reduction), feat_grad[c] += out_grad[c] * depth (in-order fmaf chain).
__device__ __forceinline__ float fbbev_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NCH>
__global__ void __launch_bounds__(256)
k_pool_bwd(int c, int n_intervals, const float* __restrict__ out_grad,
           const float* __restrict__ depth, const float* __restrict__ feat,
           const int* __restrict__ rd, const int* __restrict__ rf, const int* __restrict__ rb,
           const int* __restrict__ starts, const int* __restrict__ lengths,
           float* __restrict__ depth_grad, float* __restrict__ feat_grad) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= n_intervals) return;  // wave-uniform exit
    const int s = starts[wave], len = lengths[wave];
    const long long pf = rf[s];
    float f[NCH], g[NCH];
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
        const int ch = lane + 64 * r;
        f[r] = (ch < c) ? feat[pf * c + ch] : 0.f;
        g[r] = 0.f;
    }
    int k = 0;
    for (; k + 2 <= len; k += 2) {
        const long long pb0 = rb[s + k], pb1 = rb[s + k + 1];
        const int pd0 = rd[s + k], pd1 = rd[s + k + 1];
        const float d0 = depth[pd0], d1 = depth[pd1];
        float og0[NCH], og1[NCH];
#pragma unroll
        for (int r = 0; r < NCH; ++r) {
            const int ch = lane + 64 * r;
            og0[r] = (ch < c) ? out_grad[pb0 * c + ch] : 0.f;
            og1[r] = (ch < c) ? out_grad[pb1 * c + ch] : 0.f;
        }
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int r = 0; r < NCH; ++r) {
            p0 = fmaf(og0[r], f[r], p0);
            p1 = fmaf(og1[r], f[r], p1);
            g[r] = fmaf(og0[r], d0, g[r]);
            g[r] = fmaf(og1[r], d1, g[r]);
        }
        p0 = fbbev_wave_sum(p0);
        p1 = fbbev_wave_sum(p1);
        if (lane == 0) { depth_grad[pd0] = p0; depth_grad[pd1] = p1; }
    }
    if (k < len) {
        const long long pb0 = rb[s + k];
        const int pd0 = rd[s + k];
        const float d0 = depth[pd0];
        float p0 = 0.f;
#pragma unroll
        for (int r = 0; r < NCH; ++r) {
            const int ch = lane + 64 * r;
            const float og = (ch < c) ? out_grad[pb0 * c + ch] : 0.f;
            p0 = fmaf(og, f[r], p0);
            g[r] = fmaf(og, d0, g[r]);
        }
        p0 = fbbev_wave_sum(p0);
        if (lane == 0) depth_grad[pd0] = p0;
    }
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
        const int ch = lane + 64 * r;
        if (ch < c) feat_grad[pf * c + ch] = g[r];
    }
}
