// pool_kernels.h -- bev_pool_v2 (lift-splat voxel pooling) kernels for gfx950.
//
// Reference semantics: mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:18-45 (forward),
// :64-118 (backward).  The reference maps one CUDA thread to one (interval, channel) scalar
// (forward) and one thread to a whole feature-pixel interval (backward, 17 blocks on the shipped
// config).  Here:
//   * forward "rows"  : a group of C/4 lanes owns one interval; each lane carries 4 channels as a
//     float4, so a feature row is one fully coalesced 256-320 B read, and the interval is an
//     fmaf chain in the given sorted order (bit-identical arithmetic to the reference kernel).
//   * forward "dense" (k_pool_fwd_dense2, end of file): a 256-thread workgroup owns a tile of TV
//     consecutive voxels of one (b,z) plane for all (or half of the) C channels.  Sparse per-voxel sums are staged in an LDS tile [C][TV] (the
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

// ================================================================ fused dense forward, v2
// Same contract and same bits as k_pool_fwd_dense, restructured for memory latency:
//   * level 1: tile_istart / tile_pstart (two broadcast loads) give the tile's interval range
//     [i0,i1) AND its point range [p0,p1) -- the points of a tile are contiguous in the sorted
//     arrays -- so
//   * level 2: interval metadata (start, length, voxel) and the first NP point indices
//     (ranks_depth, ranks_feat) are staged into LDS with coalesced loads by the whole block,
//     overlapping the zero-fill of the LDS tile;
//   * level 3: lane groups gather depth scalars and feature rows (the only remaining
//     latency-exposed dependent loads) and run the in-order fmaf chains.
// NT threads per workgroup (64 = one wave per tile: no cross-wave barrier, tiles fully decoupled).
// CPL channels per lane (4 or 8): with 8, C=80 needs 10 lanes per interval -> 25 intervals per
// block in flight instead of 12.  `csplit` splits the channel range over blockIdx.y-like halves
// (tile LDS shrinks -> more resident blocks per CU).  ST selects the store cache policy.
#define FBBEV_NP_STAGE 512

template <int CPL, int U>
__device__ __forceinline__ void fbbev_interval_sum_staged(int c, int s, int len, int p0,
                                                          const int* __restrict__ prd_lds,
                                                          const int* __restrict__ prf_lds,
                                                          const float* __restrict__ depth,
                                                          const float* __restrict__ fbase,
                                                          const int* __restrict__ rd,
                                                          const int* __restrict__ rf,
                                                          float (&acc)[CPL]) {
    // U points per batch: their index / depth / feature-row loads are all issued before the first fmaf,
    // so an interval of len points costs ceil(len/U) memory round trips; the fmaf order stays k = 0,1,2,...
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    int k = 0;
    for (; k + U <= len; k += U) {
        // index pairs: ALWAYS an LDS read (clamped into the staged range), overridden from global memory only by the lanes
        // whose points lie beyond it (rare: a tile with more than FBBEV_NP_STAGE points).  Written as `idx < N ? lds[idx] :
        // global[idx]` the compiler selected between the two POINTERS and issued flat_load_dword -- eight flat loads and ~50
        // VALU instructions of 64-bit address selects per batch (tools/isa_waits.py; flat loads also count on both wait counters)
        int pd[U], pf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = s + k + u;
            const int il = idx < FBBEV_NP_STAGE ? idx : FBBEV_NP_STAGE - 1;
            pd[u] = prd_lds[il]; pf[u] = prf_lds[il];
        }
        if (s + k + U > FBBEV_NP_STAGE) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = s + k + u;
                if (idx >= FBBEV_NP_STAGE) { pd[u] = rd[p0 + idx]; pf[u] = rf[p0 + idx]; }
            }
        }
        float d[U];
        float f[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = depth[pd[u]];
            const float* fp = fbase + (long long)pf[u] * c;
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
                f[u][4 * q] = t[0]; f[u][4 * q + 1] = t[1]; f[u][4 * q + 2] = t[2]; f[u][4 * q + 3] = t[3];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[j] = fmaf(f[u][j], d[u], acc[j]);
        }
    }
    for (; k < len; ++k) {
        const int idx = s + k;
        const int il = idx < FBBEV_NP_STAGE ? idx : FBBEV_NP_STAGE - 1;
        // explicit LDS-address-space reads (see above; plain `prd_lds[il]` followed by the override was merged into a flat
        // load again): this loop takes the short intervals, i.e. most of them at 0.4 m voxels
        int pd = fbbev_lds_ld_i32(prd_lds + il), pf = fbbev_lds_ld_i32(prf_lds + il);
        if (idx >= FBBEV_NP_STAGE) { pd = rd[p0 + idx]; pf = rf[p0 + idx]; }
        const float d0 = depth[pd];
        const float* fp = fbase + (long long)pf * c;
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
            const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(t[e], d0, acc[4 * q + e]);
        }
    }
}

template <int V> struct fbbev_int_c { static constexpr int value = V; };
// The same sum with batches of EIGHT points first (then one batch of four, then single points): the knob VERDICT r4 (weak 9) left
// open for grids with long intervals (the shipped 100 x 100 x 8 grid: 4.2 points per voxel) -- half the dependent memory round
// trips per interval of >= 8 points for 2x the registers of a batch.  The fmaf order stays k = 0, 1, 2, ...: the same bits.
template <int CPL>
__device__ __forceinline__ void fbbev_interval_sum_staged8(int c, int s, int len, int p0, const int* __restrict__ prd_lds,
                                                           const int* __restrict__ prf_lds, const float* __restrict__ depth,
                                                           const float* __restrict__ fbase, const int* __restrict__ rd,
                                                           const int* __restrict__ rf, float (&acc)[CPL]) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    int k = 0;
    auto batch = [&](auto UC) {
        constexpr int U = decltype(UC)::value;
        int pd[U], pf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = s + k + u;
            const int il = idx < FBBEV_NP_STAGE ? idx : FBBEV_NP_STAGE - 1;
            pd[u] = fbbev_lds_ld_i32(prd_lds + il); pf[u] = fbbev_lds_ld_i32(prf_lds + il);
        }
        if (s + k + U > FBBEV_NP_STAGE) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = s + k + u;
                if (idx >= FBBEV_NP_STAGE) { pd[u] = rd[p0 + idx]; pf[u] = rf[p0 + idx]; }
            }
        }
        float d[U];
        float f[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = depth[pd[u]];
            const float* fp = fbase + (long long)pf[u] * c;
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
                f[u][4 * q] = t[0]; f[u][4 * q + 1] = t[1]; f[u][4 * q + 2] = t[2]; f[u][4 * q + 3] = t[3];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[j] = fmaf(f[u][j], d[u], acc[j]);
        }
        k += U;
    };
    while (k + 8 <= len) batch(fbbev_int_c<8>{});
    if (k + 4 <= len) batch(fbbev_int_c<4>{});
    for (; k < len; ++k) {
        const int idx = s + k;
        const int il = idx < FBBEV_NP_STAGE ? idx : FBBEV_NP_STAGE - 1;
        int pd = fbbev_lds_ld_i32(prd_lds + il), pf = fbbev_lds_ld_i32(prf_lds + il);
        if (idx >= FBBEV_NP_STAGE) { pd = rd[p0 + idx]; pf = rf[p0 + idx]; }
        const float d0 = depth[pd];
        const float* fp = fbase + (long long)pf * c;
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
            const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(t[e], d0, acc[4 * q + e]);
        }
    }
}

// gate[0] = build number (cache_state[1]) the tile table was built for, gate[1] = "keep the table" decision of THIS call:
// the index set is unchanged (cache_state[0] != 0) and the table belongs to that build.  A separate 1-thread launch so
// that no workgroup of the table kernel can observe the refreshed build number of its own launch.
__global__ void k_tile_table_gate(const int* __restrict__ cache_state, int* __restrict__ gate) {
    const int build = cache_state[1];
    gate[1] = (cache_state[0] != 0 && gate[0] == build) ? 1 : 0;
    gate[0] = build;
}

// tile_meta[2*t] = first interval of tile t, tile_meta[2*t+1] = first point of tile t (t in [0,n_tiles])
__global__ void __launch_bounds__(256)
k_tile_lower_bound2(int n_tiles, int tiles_per_plane, int YX, int TV,
                    const int* __restrict__ interval_rank, const int* __restrict__ starts,
                    const int* __restrict__ counts /* [P, I] */, int n_intervals_max,
                    const int* __restrict__ skip /* cached index set unchanged: keep the table */,
                    int* __restrict__ tile_meta) {
    if (skip != nullptr && *skip != 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    int P = counts[0];
    int n = counts[1];
    if (n > n_intervals_max) n = n_intervals_max;
    if (n < 0 || P < 0) { n = 0; P = 0; }          // a failed rank build reports P = I = -1: pool nothing
    const int plane = t / tiles_per_plane, k = t - plane * tiles_per_plane;
    const long long target = (long long)plane * YX + (long long)k * TV;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)interval_rank[mid] < target) lo = mid + 1; else hi = mid;
    }
    tile_meta[2 * t] = lo;
    tile_meta[2 * t + 1] = (lo < n) ? starts[lo] : P;
}

// OT: element type of `out` -- 0 f32, 1 bf16, 2 f16.  The per-voxel sums are ALWAYS the fp32 in-order fmaf chains;
// 16-bit storage rounds them once (nearest-even) at the store: BASELINE configs[1] (bf16) / configs[4] (fp16).
// IEEE binary32 -> binary16 bits, round to nearest even, overflow -> inf, NaN -> quiet NaN (integer arithmetic only,
// so the CPU emulator and the GPU produce the same bits)
__device__ __forceinline__ unsigned int fbbev_f32_to_f16(float f) {
    unsigned int u;
    __builtin_memcpy(&u, &f, 4);
    const unsigned int sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    unsigned int o;
    if (u >= ((127u + 16u) << 23)) {
        o = (u > 0x7f800000u) ? 0x7e00u : 0x7c00u;
    } else if (u < (113u << 23)) {                       // result is a half subnormal (or zero)
        const unsigned int magic = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        float t, m;
        __builtin_memcpy(&t, &u, 4);
        __builtin_memcpy(&m, &magic, 4);
        t += m;                                          // the fp32 add performs the nearest-even rounding
        unsigned int r;
        __builtin_memcpy(&r, &t, 4);
        o = r - magic;
    } else {
        const unsigned int odd = (u >> 13) & 1u;
        u += ((unsigned int)(15 - 127) << 23) + 0xfffu;
        u += odd;
        o = u >> 13;
    }
    return sign | o;
}

template <int OT>
__device__ __forceinline__ unsigned int fbbev_pack2(float lo, float hi) {
    if constexpr (OT == 1) {
        unsigned int a, b;
        __builtin_memcpy(&a, &lo, 4);
        __builtin_memcpy(&b, &hi, 4);
        a = ((a & 0x7fffffffu) > 0x7f800000u) ? ((a >> 16) | 0x40u) : ((a + 0x7fffu + ((a >> 16) & 1u)) >> 16);
        b = ((b & 0x7fffffffu) > 0x7f800000u) ? ((b >> 16) | 0x40u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
        return (a & 0xffffu) | (b << 16);
    } else {
        return fbbev_f32_to_f16(lo) | (fbbev_f32_to_f16(hi) << 16);
    }
}

// T16 (16-bit output, no addend): the LDS tile itself holds 16-bit elements.  A voxel's sum is complete when it is
// written to the tile (one lane group accumulates the whole interval in registers), so rounding it there is the same
// single rounding as rounding at the store -- but the tile is half the size: a workgroup can own twice the voxels at the
// same LDS footprint and keeps the same number of OUTPUT bytes in flight per CU as the fp32 kernel (with an fp32 tile the
// 16-bit variants were bound by the latency of the dependent-load chain at half the bytes per workgroup: 0.42 of the HBM
// peak at BASELINE configs[1], profiles/r01_bench_storage_variants.jsonl).
// DIAG (measurement only, bench.py `roofline.store_floor_ms`): 1 = the kernel's STORE pattern alone -- same grid, tile
// walk, XCD order and `sc1 nt` 16-byte stores, every tile treated as empty (no metadata, no gathers, no LDS); 2 = everything
// but the depth / feature gathers and their fmaf chains (tile metadata, interval / point-index staging, LDS tile, barriers,
// stores); 3 = everything but the stores (metadata, staging, gathers, fmaf chains, LDS tile: what the stores have to hide).
// 0 = the product kernel; the diagnostic instantiations write zeros (3: nothing) and are reachable only through
// fbbev_diag_pool_store_floor.
#define FBBEV_POOL_SPLIT_GROUPS 32     // lane groups a long interval is split over (bounds the extra LDS: 32 x CC floats)
// SPLIT > 0 (opt-in tolerance mode, FBBEV_POOL_SPLIT_LONG): an interval longer than SPLIT points is summed by ALL lane
// groups of the workgroup -- group g < 32 takes the g-th contiguous chunk of its points (chunk = ceil(len / groups) rounded up to
// the gather batch), in order, and the partial sums are added in group order: a fixed-shape, run-to-run deterministic
// reduction that differs from the reference's serial chain (bev_pool_cuda.cu:33-38) only by fp32 reassociation (<= 1e-4 of
// the sum for the path's sizes: the bar north_star states; tested) -- the default (SPLIT = 0) stays the serial chain, bit
// for bit.  What it buys: one lane group no longer serialises a 236-point (shipped grid) or 3 894-point (BASELINE
// configs[0]) interval while the other groups of the tile wait at the barrier.
template <int TV, int CPL, int ST, int NT, int OT, bool T16 = false, int DIAG = 0, int SPLIT = 0, int GU = 4>
__global__ void __launch_bounds__(NT)
k_pool_fwd_dense2(int C, int Z, int YX, int tiles_per_plane, int csplit, int n_blocks, int swizzle,
                  long long out_stride_b, long long out_stride_c,
                  const float* __restrict__ depth, const float* __restrict__ feat,
                  const int* __restrict__ rd, const int* __restrict__ rf,
                  const int* __restrict__ interval_rank, const int* __restrict__ starts,
                  const int* __restrict__ lengths, const int* __restrict__ tile_meta,
                  const float* __restrict__ addend, float* __restrict__ out) {
    static_assert(!T16 || OT != 0, "a 16-bit tile only for 16-bit output");
    static_assert(SPLIT == 0 || (!T16 && DIAG == 0), "tolerance mode: fp32 LDS tile");
    constexpr int LD = T16 ? TV + 8 : TV + 4;  // tile row pitch in ELEMENTS (16-byte aligned rows either way)
    constexpr int Q4 = TV / 4;
    const int CC = C / csplit;                 // channels handled by this block
    float* tile = fbbev_dyn_lds_f32();         // [CC][LD] f32, or [CC][LD] 16-bit when T16
    unsigned short* tile16 = reinterpret_cast<unsigned short*>(tile);
    int* ist = T16 ? reinterpret_cast<int*>(tile16 + CC * LD)
                   : reinterpret_cast<int*>(tile + CC * LD);   // [TV] interval start relative to p0
    int* iln = ist + TV;                       // [TV]
    int* ivx = iln + TV;                       // [TV] voxel offset inside the tile
    int* prd = ivx + TV;                       // [NP_STAGE]
    int* prf = prd + FBBEV_NP_STAGE;           // [NP_STAGE]
    int* lng = prf + FBBEV_NP_STAGE;           // SPLIT: [1 + TV] count + list of the tile's long intervals
    float* part = reinterpret_cast<float*>(lng + TV + 4);   // SPLIT: [FBBEV_POOL_SPLIT_GROUPS][CC] partial sums of the interval being split
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    if (swizzle) {
        // XCD-aware order: the dispatcher places block b on XCD b%8.  Chunks of S = 2^(swizzle-1)
        // consecutive tiles are dealt round-robin to the 8 XCDs, so each XCD's L2 sees runs of S
        // adjacent tiles (contiguous S*TV*4-byte spans per channel row) while the 8 XCDs stay
        // de-phased by S tiles instead of marching in lock-step 1/8 of the tensor apart.
        const int sh = swizzle - 1;
        const int xcd = bid & 7, j = bid >> 3;
        bid = ((((j >> sh) << 3) + xcd) << sh) + (j & ((1 << sh) - 1));
    }
    if (bid >= n_blocks) return;
    // tile-major: the csplit channel groups of a tile are adjacent workgroups (index / feat reads shared through L2)
    const int t_ = bid / csplit;
    const int half = bid - t_ * csplit;
    const int c0 = half * CC;
    // tile order.  Plain: plane-major (consecutive workgroups = consecutive 128-voxel runs of one plane).  With an addend (round 5): the Z
    // planes of a BEV tile are CONSECUTIVE workgroups -- they all add the same (C, TV) piece of the (B, C, Y, X) addend, which then comes
    // from the XCD's L2 most of the time instead of being re-read from far memory for every plane.  Every tile still writes its own runs.
    int plane, k;
    if (addend != nullptr) {                                                         // uniform
        const int per_b = tiles_per_plane * Z;
        const int bb = t_ / per_b, r = t_ - bb * per_b;
        k = r / Z;
        plane = bb * Z + (r - k * Z);
    } else {
        plane = t_ / tiles_per_plane; k = t_ - plane * tiles_per_plane;
    }
    const int t = plane * tiles_per_plane + k;                                      // index of the tile in tile_meta (the interval build's order)
    const int b = plane / Z, z = plane - b * Z;
    const int v0 = k * TV;
    const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
    const int i0 = DIAG == 1 ? 0 : tile_meta[2 * t], p0 = DIAG == 1 ? 0 : tile_meta[2 * t + 1];
    const int i1 = DIAG == 1 ? 0 : tile_meta[2 * t + 2], p1 = DIAG == 1 ? 0 : tile_meta[2 * t + 3];
    const long long cstride = out_stride_c;      // elements between channels (Z*YX when contiguous)
    const long long oofs = (long long)b * out_stride_b + (long long)z * YX + v0 + (long long)c0 * cstride;
    float* __restrict__ obase = out + oofs;                                         // OT == 0
    unsigned short* __restrict__ obase16 = reinterpret_cast<unsigned short*>(out) + oofs;   // OT != 0
    const int n4 = CC * Q4;
    constexpr int Q8 = TV / 8;
    // optional epilogue: out[b,c,z,y,x] = pooled + addend[b,c,y,x] (the re-add of the refined BEV, fbocc.py:365-366,
    // fused into the one pass that writes the volume); addend is (B,C,Y,X) contiguous, broadcast over z
    const float* __restrict__ ab = addend ? addend + ((long long)b * C + c0) * YX + v0 : nullptr;

    if (i0 == i1) {
        if constexpr (DIAG == 3) return;                     // gathers alone: an empty tile has none
        fbbev_v4f zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.f;
        if constexpr (OT == 0) {
            if (ab) {                                                                // uniform
                // round 5: the addend pieces of a batch are REQUESTED before the first is stored (one load, wait, store per piece before).
                // Requesting them at the top of the kernel, next to the tile metadata, costs the PLAIN kernel 8 registers and 3 % (0.766 ->
                // 0.741 of the HBM peak): not done
                constexpr int NB = 5;
                for (int idx0 = tid; idx0 < n4; idx0 += NT * NB) {
                    fbbev_v4f av[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int idx = idx0 + NT * u < n4 ? idx0 + NT * u : idx0;
                        const int c = idx / Q4, j = (idx - c * Q4) * 4;
                        av[u] = *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + (j < nv ? j : 0));
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int idx = idx0 + NT * u;
                        const int c = idx / Q4, j = (idx - c * Q4) * 4;
                        if (idx < n4 && j < nv) fbbev_store4<ST>(obase + c * cstride + j, av[u]);
                    }
                }
            } else {
                for (int idx = tid; idx < n4; idx += NT) {
                    const int c = idx / Q4, j = (idx - c * Q4) * 4;
                    if (j < nv) fbbev_store4<ST>(obase + c * cstride + j, zero);
                }
            }
        } else {
            for (int idx = tid; idx < CC * Q8; idx += NT) {
                const int c = idx / Q8, j = (idx - c * Q8) * 8;
                if (j < nv) {
                    fbbev_v4f val = zero;
                    if (ab) {
                        const fbbev_v4f lo = *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j);
                        const fbbev_v4f hi = *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j + 4);
                        unsigned int pk[4] = {fbbev_pack2<OT>(lo[0], lo[1]), fbbev_pack2<OT>(lo[2], lo[3]),
                                              fbbev_pack2<OT>(hi[0], hi[1]), fbbev_pack2<OT>(hi[2], hi[3])};
                        __builtin_memcpy(&val, pk, 16);
                    }
                    fbbev_store4<ST>(reinterpret_cast<float*>(obase16 + c * cstride + j), val);
                }
            }
        }
        return;
    }

    const int ni = i1 - i0;
    const int np = p1 - p0;
    const int rank0 = plane * YX + v0;
    if constexpr (SPLIT > 0) {
        if (tid == 0) lng[0] = 0;
        __syncthreads();
    }
    for (int j = tid; j < ni; j += NT) {
        ist[j] = starts[i0 + j] - p0;
        const int len = lengths[i0 + j];
        iln[j] = len;
        ivx[j] = interval_rank[i0 + j] - rank0;
        if constexpr (SPLIT > 0) {
            if (len > SPLIT) lng[1 + atomicAdd(&lng[0], 1)] = j;    // list order is arbitrary: every entry is summed on its own
        }
    }
    const int nps = np < FBBEV_NP_STAGE ? np : FBBEV_NP_STAGE;
    for (int j = tid; j < nps; j += NT) { prd[j] = rd[p0 + j]; prf[j] = rf[p0 + j]; }
    if constexpr (T16) {
        for (int idx = tid; idx < CC * LD / 2; idx += NT) reinterpret_cast<unsigned int*>(tile16)[idx] = 0u;
    } else {
        for (int idx = tid; idx < CC * LD; idx += NT) tile[idx] = 0.f;
    }
    __syncthreads();

    {
        const int lpi = CC / CPL;
        const int gpb = NT / lpi;
        const int g = tid / lpi, slot = tid - g * lpi;
        if (g < gpb) {
            const float* fbase = feat + c0 + slot * CPL;
            for (int i = g; i < ni; i += gpb) {
                const int v = ivx[i];
                float acc[CPL];
                if constexpr (SPLIT > 0) {
                    if (iln[i] > SPLIT) continue;               // summed by the whole workgroup below
                }
                if constexpr (DIAG == 2) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
                } else if constexpr (GU == 8) {
                    fbbev_interval_sum_staged8<CPL>(C, ist[i], iln[i], p0, prd, prf, depth, fbase, rd, rf, acc);
                } else {
                    fbbev_interval_sum_staged<CPL, 4>(C, ist[i], iln[i], p0, prd, prf, depth, fbase, rd, rf, acc);
                }
                if (v >= 0 && v < nv) {
                    if constexpr (T16) {
                        unsigned short* dst = tile16 + (slot * CPL) * LD + v;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) dst[j * LD] = (unsigned short)(fbbev_pack2<OT>(acc[j], 0.f) & 0xffffu);
                    } else {
                        float* dst = tile + (slot * CPL) * LD + v;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) dst[j * LD] = acc[j];
                    }
                }
            }
        }
        if constexpr (SPLIT > 0) {
            const int nl = lng[0];                                   // written before the first barrier: block-uniform
            const int ngs = gpb < FBBEV_POOL_SPLIT_GROUPS ? gpb : FBBEV_POOL_SPLIT_GROUPS;   // lane groups that share an interval
            const float* fbase = feat + c0 + slot * CPL;
            for (int q = 0; q < nl; ++q) {
                const int i = lng[1 + q];
                const int len = iln[i], s0 = ist[i], v = ivx[i];
                const int chunk = (((len + ngs - 1) / ngs) + 3) & ~3;   // whole gather batches per group
                if (g < ngs) {
                    const int sub = g * chunk;
                    int sl = len - sub;
                    sl = sl < 0 ? 0 : (sl > chunk ? chunk : sl);
                    float acc[CPL];
                    fbbev_interval_sum_staged<CPL, 4>(C, s0 + sub, sl, p0, prd, prf, depth, fbase, rd, rf, acc);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) part[g * CC + slot * CPL + j] = acc[j];
                }
                __syncthreads();
                if (tid < CC) {                                      // partial sums in group order: a fixed shape
                    float sum = part[tid];
                    for (int gg = 1; gg < ngs; ++gg) sum += part[gg * CC + tid];
                    if (v >= 0 && v < nv) tile[tid * LD + v] = sum;
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if constexpr (DIAG == 3) return;                         // (the LDS tile writes keep the gathers alive)

    if constexpr (OT == 0) {
        if (ab) {                                                                    // uniform; batches as in the empty-tile branch
            constexpr int NB = 5;
            for (int idx0 = tid; idx0 < n4; idx0 += NT * NB) {
                fbbev_v4f av[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int idx = idx0 + NT * u < n4 ? idx0 + NT * u : idx0;
                    const int c = idx / Q4, j = (idx - c * Q4) * 4;
                    av[u] = *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + (j < nv ? j : 0));
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int idx = idx0 + NT * u;
                    const int c = idx / Q4, j = (idx - c * Q4) * 4;
                    if (idx < n4 && j < nv) {
                        fbbev_v4f val = *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j);
                        val += av[u];
                        fbbev_store4<ST>(obase + c * cstride + j, val);
                    }
                }
            }
        } else {
            for (int idx = tid; idx < n4; idx += NT) {
                const int c = idx / Q4, j = (idx - c * Q4) * 4;
                if (j < nv) fbbev_store4<ST>(obase + c * cstride + j, *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j));
            }
        }
    } else if constexpr (T16) {
        for (int idx = tid; idx < CC * Q8; idx += NT) {
            const int c = idx / Q8, j = (idx - c * Q8) * 8;
            if (j < nv)
                fbbev_store4<ST>(reinterpret_cast<float*>(obase16 + c * cstride + j),
                                 *reinterpret_cast<const fbbev_v4f*>(tile16 + c * LD + j));
        }
    } else {
        for (int idx = tid; idx < CC * Q8; idx += NT) {
            const int c = idx / Q8, j = (idx - c * Q8) * 8;
            if (j < nv) {
                fbbev_v4f lo = *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j);
                fbbev_v4f hi = *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j + 4);
                if (ab) {
                    lo += *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j);
                    hi += *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j + 4);
                }
                unsigned int pk[4] = {fbbev_pack2<OT>(lo[0], lo[1]), fbbev_pack2<OT>(lo[2], lo[3]),
                                      fbbev_pack2<OT>(hi[0], hi[1]), fbbev_pack2<OT>(hi[2], hi[3])};
                fbbev_v4f val;
                __builtin_memcpy(&val, pk, 16);
                fbbev_store4<ST>(reinterpret_cast<float*>(obase16 + c * cstride + j), val);
            }
        }
    }
}

// ================================================================ dense forward, software-pipelined over a run of tiles (round 4)
// k_pool_fwd_dense2 walks a five-level dependent chain per tile -- tile table -> interval metadata / point indices -> (barrier)
// -> depth / feature gathers -> (barrier) -> stores -- with nothing of the next tile in flight: at the shipped grid (100x100x8,
// 68 % of the voxels occupied, ~87 intervals x 4 points per 128-voxel tile) the waves are parked 61 % of the time and the
// kernel reaches 0.35 of the HBM peak where the sparse BASELINE configs[1] grid reaches 0.76 (profiles/r03_scope_table.json,
// r02_pmc_forward_REF_B16.json).  Here a workgroup owns `tpw` CONSECUTIVE tiles of one channel group and double-buffers the
// staging arrays: while tile t's gathers run, the interval metadata and the first FBBEV_NP_STAGE point indices of tile t + 1
// are already in flight (registers -> the other LDS buffer after the store phase), and its tile-table words were fetched
// one tile earlier still.  The store phase re-zeroes exactly the tile elements the same thread just read, so the LDS tile
// needs no separate clear and the loop keeps two barriers per tile.  Same per-interval fmaf chains (fbbev_interval_sum_staged)
// => the same bits as k_pool_fwd_dense2; fp32 volume, optional re-add epilogue.
template <int TV, int CPL, int ST, int NT>
__global__ void __launch_bounds__(NT)
k_pool_fwd_dense_pipe(int C, int Z, int YX, int tiles_per_plane, int csplit, int n_blocks, int swizzle, int tpw, int n_tiles,
                      long long out_stride_b, long long out_stride_c,
                      const float* __restrict__ depth, const float* __restrict__ feat,
                      const int* __restrict__ rd, const int* __restrict__ rf,
                      const int* __restrict__ interval_rank, const int* __restrict__ starts,
                      const int* __restrict__ lengths, const int* __restrict__ tile_meta,
                      const float* __restrict__ addend, float* __restrict__ out) {
    constexpr int LD = TV + 4, Q4 = TV / 4;
    constexpr int STG = 3 * TV + 2 * FBBEV_NP_STAGE;                       // ints of one staging buffer
    constexpr int IPT = (TV + NT - 1) / NT, PPT = (FBBEV_NP_STAGE + NT - 1) / NT;
    const int CC = C / csplit;
    float* tile = fbbev_dyn_lds_f32();                                      // [CC][LD]
    int* stg = reinterpret_cast<int*>(tile + CC * LD);                      // [2][STG]: ist | iln | ivx | prd | prf
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    if (swizzle) {                                                          // as k_pool_fwd_dense2, on runs of tiles
        const int sh = swizzle - 1;
        const int xcd = bid & 7, j = bid >> 3;
        bid = ((((j >> sh) << 3) + xcd) << sh) + (j & ((1 << sh) - 1));
    }
    if (bid >= n_blocks) return;
    const int grp = bid / csplit, half = bid - grp * csplit;
    const int c0 = half * CC;
    const int t0 = grp * tpw, t1 = t0 + tpw < n_tiles ? t0 + tpw : n_tiles;
    const long long cstride = out_stride_c;
    const int n4 = CC * Q4;
    // the tile table words of tiles t (a), t + 1 (b), t + 2 (c): (first interval, first point); entry n_tiles closes the table
    int ia = tile_meta[2 * t0], pa = tile_meta[2 * t0 + 1];
    int ib = tile_meta[2 * (t0 + 1)], pb = tile_meta[2 * (t0 + 1) + 1];
    int r_st[IPT], r_ln[IPT], r_vx[IPT], r_rd[PPT], r_rf[PPT];
    auto issue = [&](int i0, int p0, int i1, int p1) {                      // staging loads of one tile into registers
        const int ni = i1 - i0, np = p1 - p0;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int j = tid + q * NT;
            const int jj = i0 + (j < ni ? j : 0);                           // clamped: lanes beyond the tile load a duplicate
            r_st[q] = starts[jj]; r_ln[q] = lengths[jj]; r_vx[q] = interval_rank[jj];
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int j = tid + q * NT;
            const int jj = p0 + (j < np ? j : 0);
            r_rd[q] = rd[jj]; r_rf[q] = rf[jj];
        }
    };
    auto commit = [&](int buf, int i0, int p0, int i1, int p1, int rank0) {  // registers -> staging buffer `buf`
        int* ist = stg + buf * STG;
        int *iln = ist + TV, *ivx = iln + TV, *prd = ivx + TV, *prf = prd + FBBEV_NP_STAGE;
        const int ni = i1 - i0, np = p1 - p0;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int j = tid + q * NT;
            if (j < ni) { ist[j] = r_st[q] - p0; iln[j] = r_ln[q]; ivx[j] = r_vx[q] - rank0; }
        }
        const int nps = np < FBBEV_NP_STAGE ? np : FBBEV_NP_STAGE;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int j = tid + q * NT;
            if (j < nps) { prd[j] = r_rd[q]; prf[j] = r_rf[q]; }
        }
    };
    auto rank_of = [&](int t) { const int plane = t / tiles_per_plane; return plane * YX + (t - plane * tiles_per_plane) * TV; };
    if (ib > ia) issue(ia, pa, ib, pb);
    for (int idx = tid; idx < CC * LD; idx += NT) tile[idx] = 0.f;          // once: the store phase re-zeroes what it reads
    if (ib > ia) commit(0, ia, pa, ib, pb, rank_of(t0));
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        const bool has_next = t + 1 < t1;
        // table words of tile t + 2's start = tile t + 1's end (clamped to the closing entry)
        const int tn = t + 2 <= n_tiles ? t + 2 : n_tiles;
        const int ic = tile_meta[2 * tn], pc = tile_meta[2 * tn + 1];
        const bool next_full = has_next && ic > ib;
        if (next_full) issue(ib, pb, ic, pc);                               // in flight under this tile's gathers
        const int plane = t / tiles_per_plane, k = t - plane * tiles_per_plane;
        const int b = plane / Z, z = plane - b * Z;
        const int v0 = k * TV;
        const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
        const int ni = ib - ia;
        const long long oofs = (long long)b * out_stride_b + (long long)z * YX + v0 + (long long)c0 * cstride;
        float* __restrict__ obase = out + oofs;
        const float* __restrict__ ab = addend ? addend + ((long long)b * C + c0) * YX + v0 : nullptr;
        if (ni == 0) {                                                      // block-uniform: an empty tile is zeros (+ addend)
            fbbev_v4f zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.f;
            for (int idx = tid; idx < n4; idx += NT) {
                const int c = idx / Q4, j = (idx - c * Q4) * 4;
                if (j < nv) fbbev_store4<ST>(obase + c * cstride + j,
                                             ab ? *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j) : zero);
            }
        } else {
            const int* ist = stg + cur * STG;
            const int *iln = ist + TV, *ivx = iln + TV, *prd = ivx + TV, *prf = prd + FBBEV_NP_STAGE;
            __syncthreads();                                                // staging(cur) committed, tile zeroed
            {
                const int lpi = CC / CPL;
                const int gpb = NT / lpi;
                const int g = tid / lpi, slot = tid - g * lpi;
                if (g < gpb) {
                    const float* fbase = feat + c0 + slot * CPL;
                    for (int i = g; i < ni; i += gpb) {
                        const int v = ivx[i];
                        float acc[CPL];
                        fbbev_interval_sum_staged<CPL, 4>(C, ist[i], iln[i], pa, prd, prf, depth, fbase, rd, rf, acc);
                        if (v >= 0 && v < nv) {
                            float* dst = tile + (slot * CPL) * LD + v;
#pragma unroll
                            for (int j = 0; j < CPL; ++j) dst[j * LD] = acc[j];
                        }
                    }
                }
            }
            __syncthreads();
            for (int idx = tid; idx < n4; idx += NT) {
                const int c = idx / Q4, j = (idx - c * Q4) * 4;
                float* tp = tile + c * LD + j;
                fbbev_v4f val = *reinterpret_cast<const fbbev_v4f*>(tp);
                fbbev_v4f zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.f;
                *reinterpret_cast<fbbev_v4f*>(tp) = zero;                   // the next tile starts from a clear tile
                if (j < nv) {
                    if (ab) val += *reinterpret_cast<const fbbev_v4f*>(ab + (long long)c * YX + j);
                    fbbev_store4<ST>(obase + c * cstride + j, val);
                }
            }
        }
        if (next_full) commit(cur ^ 1, ib, pb, ic, pc, rank_of(t + 1));     // buffer cur ^ 1 was last read before this tile's barriers
        ia = ib; pa = pb; ib = ic; pb = pc;
    }
}

__host__ __device__ inline long long gridDim_stride(int n_blocks, int csplit, int tiles_per_plane) { return n_blocks / (csplit * tiles_per_plane); }   // = B

// ================================================================ Z-mean of the pooled volume without the volume
// lss_bev = bev_feat.mean(-1) (fbocc.py:359: the backward projection's input) computed straight from the index
// tensors: a workgroup owns TV consecutive (y,x) voxels x CC channels and walks the Z planes in ascending order,
// accumulating each plane's per-voxel sums (the same in-order fmaf chains) into one LDS tile; out (B,C,Y,X) =
// tile / Z.  The reference writes the (B,C,Z,Y,X) volume, permutes it and reads it back for this mean; with this
// kernel before and the `addend` epilogue of k_pool_fwd_dense2 after the backward projection the volume is written
// exactly once and never re-read.  Uses the same tile index as the dense kernel (same tile_voxels).
template <int TV, int CPL, int NT>
__global__ void __launch_bounds__(NT)
k_pool_zmean(int C, int Z, int YX, int tiles_per_plane, int csplit, int n_blocks,
             const float* __restrict__ depth, const float* __restrict__ feat,
             const int* __restrict__ rd, const int* __restrict__ rf,
             const int* __restrict__ interval_rank, const int* __restrict__ starts,
             const int* __restrict__ lengths, const int* __restrict__ tile_meta, float* __restrict__ out, int z_groups,
             float* __restrict__ partial, int rows_out, const float* __restrict__ row_bias) {
    // z_groups > 1 (round 4): the Z planes of a tile are walked ONE AFTER THE OTHER (two barriers and two dependent memory round
    // trips per plane), so with few tiles -- the shipped grid at B = 1 has 157 -- the launch is one 8-plane latency chain per CU and
    // was the largest kernel of the shipped-shape forward+backward projection (106 us).  Then workgroup (tile, zg) takes the planes
    // [zg * ceil(Z / z_groups), ...) and writes its raw sums to partial[zg][b][c][y][x]; k_pool_zmean_reduce adds the groups in
    // order and divides by Z.  (z_groups == 1: the original single pass; the association of the z sum differs between the two.)
    constexpr int LD = TV + 4;
    constexpr int Q4 = TV / 4;
    const int CC = C / csplit;
    float* tile = fbbev_dyn_lds_f32();         // [CC][LD]
    int* ist = reinterpret_cast<int*>(tile + CC * LD);
    int* iln = ist + TV;
    int* ivx = iln + TV;
    int* prd = ivx + TV;
    int* prf = prd + FBBEV_NP_STAGE;
    const int tid = threadIdx.x;
    const int bid0 = blockIdx.x;
    if (bid0 >= n_blocks * z_groups) return;
    const int zg = bid0 / n_blocks, bid = bid0 - zg * n_blocks;
    const int tk = bid / csplit, half = bid - tk * csplit;      // tk = b * tiles_per_plane + k
    const int c0 = half * CC;
    const int b = tk / tiles_per_plane, k = tk - b * tiles_per_plane;
    const int v0 = k * TV;
    const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
    const int zper = (Z + z_groups - 1) / z_groups;
    const int z_lo = zg * zper, z_hi = z_lo + zper < Z ? z_lo + zper : Z;
    for (int idx = tid; idx < CC * LD; idx += NT) tile[idx] = 0.f;
    // Round 5: the plane walk as a two-stage pipeline.  A plane costs a chain tile metadata -> interval records / point pairs -> barrier
    // -> gathers; the metadata of plane z + 2 (scalar loads) and the records of plane z + 1 (five registers per thread: its first interval
    // record and point-index pair) are requested BEFORE the gathers of plane z and land under them.  Same staging contents, same gather
    // order: identical bits.
    auto load_meta = [&](int z, int (&m)[4]) {
        const long long t = (long long)(b * Z + z) * tiles_per_plane + k;
        m[0] = tile_meta[2 * t]; m[1] = tile_meta[2 * t + 1]; m[2] = tile_meta[2 * t + 2]; m[3] = tile_meta[2 * t + 3];
    };
    int sa = 0, sl = 0, sr = 0, sd = 0, sf = 0;                   // the staged-ahead record / pair of the NEXT plane to be committed
    auto request = [&](const int (&m)[4]) {                      // (clamped, unconditional loads: one round trip)
        const int ni = m[2] - m[0], np = m[3] - m[1];
        if (ni == 0) return;                                     // block-uniform
        const int nps = np < FBBEV_NP_STAGE ? np : FBBEV_NP_STAGE;
        const int ji = tid < ni ? tid : 0, jp = tid < nps ? tid : 0;
        sa = starts[m[0] + ji]; sl = lengths[m[0] + ji]; sr = interval_rank[m[0] + ji];
        sd = rd[m[1] + jp]; sf = rf[m[1] + jp];
    };
    int m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
    if (z_lo < z_hi) load_meta(z_lo, m0);
    if (z_lo + 1 < z_hi) load_meta(z_lo + 1, m1);
    request(m0);
    for (int z = z_lo; z < z_hi; ++z) {
        if (z + 2 < z_hi) load_meta(z + 2, m2);
        const int plane = b * Z + z;
        const int i0 = m0[0], p0 = m0[1], i1 = m0[2], p1 = m0[3];
        const int ni = i1 - i0, np = p1 - p0;
        const int rank0 = plane * YX + v0;
        const int nps = np < FBBEV_NP_STAGE ? np : FBBEV_NP_STAGE;
        if (ni != 0) {                          // block-uniform
            __syncthreads();                    // previous plane's gathers are done with the staging buffers / tile init
            if (tid < ni) { ist[tid] = sa - p0; iln[tid] = sl; ivx[tid] = sr - rank0; }
            if (tid < nps) { prd[tid] = sd; prf[tid] = sf; }
            for (int j = tid + NT; j < ni; j += NT) {
                ist[j] = starts[i0 + j] - p0;
                iln[j] = lengths[i0 + j];
                ivx[j] = interval_rank[i0 + j] - rank0;
            }
            for (int j = tid + NT; j < nps; j += NT) { prd[j] = rd[p0 + j]; prf[j] = rf[p0 + j]; }
            __syncthreads();
        }
        if (z + 1 < z_hi) request(m1);          // lands under this plane's gathers
        if (ni != 0) {
            const int lpi = CC / CPL;
            const int gpb = NT / lpi;
            const int g = tid / lpi, slot = tid - g * lpi;
            if (g < gpb) {
                const float* fbase = feat + c0 + slot * CPL;
                for (int i = g; i < ni; i += gpb) {          // a voxel appears in at most one interval of a plane:
                    const int v = ivx[i];                    // its LDS cell is owned by one lane group per plane
                    float acc[CPL];
                    fbbev_interval_sum_staged<CPL, 4>(C, ist[i], iln[i], p0, prd, prf, depth, fbase, rd, rf, acc);
                    if (v >= 0 && v < nv) {
                        float* dst = tile + (slot * CPL) * LD + v;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) dst[j * LD] += acc[j];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { m0[q] = m1[q]; m1[q] = m2[q]; }
    }
    __syncthreads();
    if (rows_out) {
        // round 6 (z_groups == 1): out is (B, Y*X, C) ROWS -- the backward projection's query layout -- plus row_bias (Y*X, C) when given
        // (its bev_embedding: the same single fp32 add as the transposing pass that used to follow, fbbev_tokens_from_nchw): mean / Z + bias
        const float zr = (float)Z;
        const int cq_n = CC / 4;
        float* __restrict__ orow = out + ((long long)b * YX + v0) * C + c0;
        const float* __restrict__ brow = row_bias ? row_bias + (long long)v0 * C + c0 : nullptr;
        for (int idx = tid; idx < nv * cq_n; idx += NT) {
            const int v = idx / cq_n, cq = idx - v * cq_n;
            fbbev_v4f val;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = tile[(4 * cq + e) * LD + v] / zr;
            if (brow) {
                const fbbev_v4f bb = *reinterpret_cast<const fbbev_v4f*>(brow + (long long)v * C + 4 * cq);
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = fbbev_add(val[e], bb[e]);
            }
            fbbev_st(reinterpret_cast<fbbev_v4f*>(orow + (long long)v * C + 4 * cq), val);
        }
        return;
    }
    const float zf = z_groups > 1 ? 1.f : (float)Z;
    float* __restrict__ ob = (z_groups > 1 ? partial + (long long)zg * gridDim_stride(n_blocks, csplit, tiles_per_plane) * C * YX : out) +
                             ((long long)b * C + c0) * YX + v0;
    for (int idx = tid; idx < CC * Q4; idx += NT) {
        const int c = idx / Q4, j = (idx - c * Q4) * 4;
        if (j < nv) {
            fbbev_v4f val = *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j);
            val[0] /= zf; val[1] /= zf; val[2] /= zf; val[3] /= zf;
            fbbev_st(reinterpret_cast<fbbev_v4f*>(ob + (long long)c * YX + j), val);
        }
    }
}

// ---------------------------------------------------------------- the same Z-mean, a pixel COLUMN at a time (round 5)
// k_pool_zmean walks the Z planes of its tile one after the other: per plane a dependent chain tile metadata -> interval metadata /
// point indices -> barrier -> gathers -> barrier, i.e. ~4 memory round trips and 2 barriers times Z (BASELINE configs[2] grid, B = 4:
// 104 us for 100 MB, waves parked 82 %).  Here the metadata of ALL planes of the tile is fetched at once (Z <= 64 lanes), the
// intervals of the whole column are scattered into a dense (plane, pixel) slot table in LDS and the point indices of the whole
// column are staged in one flattened pass; then a lane group OWNS a pixel and walks its Z voxels in ascending order, adding each
// voxel's in-order fmaf chain to the pixel's running sum -- the summation order of k_pool_zmean (identical bits), 5 barriers and ~3
// dependent round trips before the gathers instead of ~4 Z.
#define FBBEV_ZC_CAP 1024                              // point-index pairs of a column staged in LDS (the rest is read from global memory)
__host__ __device__ inline size_t fbbev_zmean_col_lds_bytes(int CC, int TV, int Z) {
    return ((size_t)CC * (TV + 4) + 2 * (size_t)Z * TV + 2 * FBBEV_ZC_CAP + 6 * (size_t)(Z + 1)) * 4;
}

template <int CPL, int U>
__device__ __forceinline__ void fbbev_interval_sum_staged_n(int c, int s, int len, int p0, const int* __restrict__ prd_lds,
                                                            const int* __restrict__ prf_lds, int n_staged,
                                                            const float* __restrict__ depth, const float* __restrict__ fbase,
                                                            const int* __restrict__ rd, const int* __restrict__ rf, float (&acc)[CPL]) {
    // fbbev_interval_sum_staged with a run-time number of staged pairs (n_staged >= 0; 0 = everything from global memory)
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    const int last = n_staged > 0 ? n_staged - 1 : 0;
    int k = 0;
    for (; k + U <= len; k += U) {
        int pd[U], pf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = s + k + u;
            const int il = idx < n_staged ? idx : last;
            pd[u] = fbbev_lds_ld_i32(prd_lds + il); pf[u] = fbbev_lds_ld_i32(prf_lds + il);
        }
        if (s + k + U > n_staged) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = s + k + u;
                if (idx >= n_staged) { pd[u] = rd[p0 + idx]; pf[u] = rf[p0 + idx]; }
            }
        }
        float d[U];
        float f[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = depth[pd[u]];
            const float* fp = fbase + (long long)pf[u] * c;
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
                f[u][4 * q] = t[0]; f[u][4 * q + 1] = t[1]; f[u][4 * q + 2] = t[2]; f[u][4 * q + 3] = t[3];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[j] = fmaf(f[u][j], d[u], acc[j]);
        }
    }
    for (; k < len; ++k) {
        const int idx = s + k;
        const int il = idx < n_staged ? idx : last;
        int pd = fbbev_lds_ld_i32(prd_lds + il), pf = fbbev_lds_ld_i32(prf_lds + il);
        if (idx >= n_staged) { pd = rd[p0 + idx]; pf = rf[p0 + idx]; }
        const float d0 = depth[pd];
        const float* fp = fbase + (long long)pf * c;
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
            const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(fp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(t[e], d0, acc[4 * q + e]);
        }
    }
}

template <int TV, int CPL, int NT>
__global__ void __launch_bounds__(NT)
k_pool_zmean_col(int C, int Z, int YX, int tiles_per_plane, int csplit, int n_blocks,
                 const float* __restrict__ depth, const float* __restrict__ feat,
                 const int* __restrict__ rd, const int* __restrict__ rf,
                 const int* __restrict__ interval_rank, const int* __restrict__ starts,
                 const int* __restrict__ lengths, const int* __restrict__ tile_meta, float* __restrict__ out) {
    constexpr int LD = TV + 4, Q4 = TV / 4;
    const int CC = C / csplit;
    float* tile = fbbev_dyn_lds_f32();                           // [CC][LD] running sums of the column's pixels
    int* slot_s = reinterpret_cast<int*>(tile + CC * LD);        // [Z][TV] first point of the voxel's interval
    int* slot_l = slot_s + Z * TV;                               // [Z][TV] its length, 0 = empty voxel
    int* prd = slot_l + Z * TV;                                  // [CAP] staged point indices of the column, plane after plane
    int* prf = prd + FBBEV_ZC_CAP;
    int* zi0 = prf + FBBEV_ZC_CAP;                               // per plane: first interval | first point | exclusive prefixes of the
    int* zp0 = zi0 + (Z + 1);                                    // interval / point counts ([Z] = totals)
    int* zib = zp0 + (Z + 1);
    int* zpb = zib + (Z + 1);
    int* zni = zpb + (Z + 1);
    int* znp = zni + (Z + 1);
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    if (bid >= n_blocks) return;
    const int tk = bid / csplit, half = bid - tk * csplit;       // tk = b * tiles_per_plane + k
    const int c0 = half * CC;
    const int b = tk / tiles_per_plane, k = tk - b * tiles_per_plane;
    const int v0 = k * TV;
    const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
    // P0: tile metadata of every plane (one lane per plane), cleared sums and slots
    if (tid < Z) {
        const int t = (b * Z + tid) * tiles_per_plane + k;
        const int i0 = tile_meta[2 * t], p0 = tile_meta[2 * t + 1], i1 = tile_meta[2 * t + 2], p1 = tile_meta[2 * t + 3];
        zi0[tid] = i0; zp0[tid] = p0; zni[tid] = i1 - i0; znp[tid] = p1 - p0;
    }
    for (int idx = tid; idx < CC * LD; idx += NT) tile[idx] = 0.f;
    for (int idx = tid; idx < Z * TV; idx += NT) slot_l[idx] = 0;
    __syncthreads();
    // P1: exclusive prefixes over the planes (lane z sums the counts below it: Z <= 64 reads, all lanes in parallel)
    if (tid <= Z) {
        int a = 0;
        for (int z = 0; z < tid; ++z) a += zni[z];
        zib[tid] = a;
    } else if (tid >= 128 && tid - 128 <= Z) {                   // (Z <= 64 < 128 <= NT - 65: another wave, the same moment)
        int a = 0;
        for (int z = 0; z < tid - 128; ++z) a += znp[z];
        zpb[tid - 128] = a;
    }
    __syncthreads();
    // P2: every interval of the column into its (plane, pixel) slot; the column's point indices into the staging arrays
    {
        const int ti = zib[Z];
        for (int f = tid; f < ti; f += NT) {
            int lo = 0, hi = Z - 1;                              // the plane of flattened interval f: last z with zib[z] <= f
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (zib[mid] <= f) lo = mid; else hi = mid - 1; }
            const int i = zi0[lo] + (f - zib[lo]);
            const int v = interval_rank[i] - ((b * Z + lo) * YX + v0);
            const int st = starts[i], ln = lengths[i];
            if (v >= 0 && v < nv) { slot_s[lo * TV + v] = st; slot_l[lo * TV + v] = ln; }
        }
        int tp = zpb[Z];
        if (tp > FBBEV_ZC_CAP) tp = FBBEV_ZC_CAP;
        for (int f = tid; f < tp; f += NT) {
            int lo = 0, hi = Z - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (zpb[mid] <= f) lo = mid; else hi = mid - 1; }
            const int p = zp0[lo] + (f - zpb[lo]);
            prd[f] = rd[p]; prf[f] = rf[p];
        }
    }
    __syncthreads();
    // P3: a lane group owns a pixel and walks its voxels in ascending z (the order of k_pool_zmean: the same bits)
    {
        const int lpi = CC / CPL;
        const int gpb = NT / lpi;
        const int g = tid / lpi, slot = tid - g * lpi;
        if (g < gpb) {
            const float* fbase = feat + c0 + slot * CPL;
            for (int v = g; v < nv; v += gpb) {
                float sum[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) sum[j] = 0.f;
                bool any = false;
                for (int z = 0; z < Z; ++z) {
                    const int len = fbbev_lds_ld_i32(slot_l + z * TV + v);
                    if (len == 0) continue;
                    const int p0 = zp0[z], pb = zpb[z];
                    int nst = FBBEV_ZC_CAP - pb;
                    const int np = znp[z];
                    nst = nst < 0 ? 0 : (nst > np ? np : nst);
                    float acc[CPL];
                    fbbev_interval_sum_staged_n<CPL, 4>(C, fbbev_lds_ld_i32(slot_s + z * TV + v) - p0, len, p0, prd + (nst > 0 ? pb : 0),
                                                        prf + (nst > 0 ? pb : 0), nst, depth, fbase, rd, rf, acc);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) sum[j] += acc[j];
                    any = true;
                }
                if (any) {
                    float* dst = tile + (slot * CPL) * LD + v;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) dst[j * LD] = sum[j];
                }
            }
        }
    }
    __syncthreads();
    const float zf = (float)Z;
    float* __restrict__ ob = out + ((long long)b * C + c0) * YX + v0;
    for (int idx = tid; idx < CC * Q4; idx += NT) {
        const int c = idx / Q4, j = (idx - c * Q4) * 4;
        if (j < nv) {
            fbbev_v4f val = *reinterpret_cast<const fbbev_v4f*>(tile + c * LD + j);
            val[0] /= zf; val[1] /= zf; val[2] /= zf; val[3] /= zf;
            fbbev_st(reinterpret_cast<fbbev_v4f*>(ob + (long long)c * YX + j), val);
        }
    }
}

// out[i] = (partial[0][i] + partial[1][i] + ... in group order) / Z over n = B*C*Y*X floats (n % 4 == 0)
__global__ void __launch_bounds__(256)
k_pool_zmean_reduce(const float* __restrict__ partial, long long n, int z_groups, float zf, float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    fbbev_v4f acc = *reinterpret_cast<const fbbev_v4f*>(partial + i);
    for (int g = 1; g < z_groups; ++g) acc += *reinterpret_cast<const fbbev_v4f*>(partial + (long long)g * n + i);
    acc[0] /= zf; acc[1] /= zf; acc[2] /= zf; acc[3] /= zf;
    fbbev_st(reinterpret_cast<fbbev_v4f*>(out + i), acc);
}

// out[bc][i] = (vol[bc][0][i] + vol[bc][1][i] + ... in z order) / divisor over a (B*C, Z, Y*X) volume: the Z-mean of the training
// path (fbocc.py:359 on a materialised volume) and the Z-sum its re-add's backward needs (fbocc.py:365-366), as one HBM-bound
// pass -- the ATen reductions over the strided last dimension of the (B,C,Y,X,Z) VIEW ran at 0.9 TB/s.  YX % 4 == 0.
__global__ void __launch_bounds__(256)
k_volume_zreduce(const float* __restrict__ vol, long long n_bc, int Z, long long YX, float divisor, float* __restrict__ out) {
    const long long q = YX >> 2, i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bc * q) return;
    const long long bc = i / q, j = (i - bc * q) * 4;
    const float* src = vol + bc * (long long)Z * YX + j;
    fbbev_v4f acc = *reinterpret_cast<const fbbev_v4f*>(src);
    for (int z = 1; z < Z; ++z) acc += *reinterpret_cast<const fbbev_v4f*>(src + (long long)z * YX);
    acc[0] /= divisor; acc[1] /= divisor; acc[2] /= divisor; acc[3] /= divisor;
    *reinterpret_cast<fbbev_v4f*>(out + bc * YX + j) = acc;
}

// The same reduction for a Z-INNERMOST volume (B*C, Y*X, Z) -- an upstream gradient that arrives contiguous in the module's
// (B,C,Y,X,Z) output shape: a thread sums the Z contiguous floats of one pillar.  Z % 4 == 0.
__global__ void __launch_bounds__(256)
k_volume_zreduce_inner(const float* __restrict__ vol, long long n_pillars, int Z, float divisor, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pillars) return;
    const float* src = vol + i * Z;
    float acc = 0.f;
    for (int z = 0; z < Z; z += 4) {
        const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(src + z);
        acc += t[0]; acc += t[1]; acc += t[2]; acc += t[3];
    }
    out[i] = acc / divisor;
}

// (B*C, Y*X, Z) -> (B*C, Z, Y*X): the re-layout of such a gradient for the pooling backward (which reads x-runs of one (z, y)): a
// thread reads the Z contiguous floats of a pillar and writes them to Z planes, coalesced across the lanes (the ATen strided copy
// ran at 1.5 TB/s).  Z % 4 == 0.
__global__ void __launch_bounds__(256)
k_volume_z_to_front(const float* __restrict__ src, long long n_bc, int Z, long long YX, float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bc * YX) return;
    const long long bc = i / YX, j = i - bc * YX;
    const float* s = src + i * Z;
    float* d = dst + bc * (long long)Z * YX + j;
    for (int z = 0; z < Z; z += 4) {
        const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(s + z);
        d[(long long)z * YX] = t[0]; d[(long long)(z + 1) * YX] = t[1]; d[(long long)(z + 2) * YX] = t[2]; d[(long long)(z + 3) * YX] = t[3];
    }
}

// ================================================================ fused dense forward, channels-last
// out (B,Z,Y,X,C) -- the reference op's own output layout (QuickCumsumCuda.forward, bev_pool.py:24-38) --
// but with EVERY voxel row written exactly once (zeros for empty voxels): replaces new_zeros + kernel.
// A tile is TV consecutive voxels of the flat (B*Z*Y*X) rank space = ONE contiguous TV*C*4-byte span, so
// the store stream of the whole grid is linear like a memset (the (B,C,Z,Y,X) variant above writes C
// separate planes per tile and floors ~15 % below memset speed, profiles/r01_exp_pool_chanmajor.jsonl).
// No LDS value tile: a lane group of C/CPL lanes owns a voxel row and stores its CPL channels directly;
// LDS only holds the staged interval metadata, the voxel->interval slot map and the staged point indices.
template <int TV, int CPL, int ST, int NT>
__global__ void __launch_bounds__(NT)
k_pool_fwd_dense_cl(int C, int n_voxels, int n_blocks, int swizzle, const float* __restrict__ depth,
                    const float* __restrict__ feat, const int* __restrict__ rd, const int* __restrict__ rf,
                    const int* __restrict__ interval_rank, const int* __restrict__ starts,
                    const int* __restrict__ lengths, const int* __restrict__ tile_meta,
                    float* __restrict__ out) {
    int* ist = reinterpret_cast<int*>(fbbev_dyn_lds_f32());   // [TV]
    int* iln = ist + TV;                                      // [TV]
    int* slot = iln + TV;                                     // [TV] interval (tile-local) of each voxel, -1 = empty
    int* prd = slot + TV;                                     // [NP_STAGE]
    int* prf = prd + FBBEV_NP_STAGE;                          // [NP_STAGE]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    if (swizzle) {
        const int sh = swizzle - 1;
        const int xcd = t & 7, j = t >> 3;
        t = ((((j >> sh) << 3) + xcd) << sh) + (j & ((1 << sh) - 1));
    }
    if (t >= n_blocks) return;
    const int v0 = t * TV;
    const int nv = (n_voxels - v0 < TV) ? (n_voxels - v0) : TV;
    const int i0 = tile_meta[2 * t], p0 = tile_meta[2 * t + 1];
    const int i1 = tile_meta[2 * t + 2], p1 = tile_meta[2 * t + 3];
    float* __restrict__ obase = out + (long long)v0 * C;
    fbbev_v4f zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.f;

    if (i0 == i1) {  // empty tile: one linear run of zeros
        const int n4 = nv * (C >> 2);
        for (int idx = tid; idx < n4; idx += NT) fbbev_store4<ST>(obase + 4 * idx, zero);
        return;
    }

    const int ni = i1 - i0;
    const int np = p1 - p0;
    for (int j = tid; j < TV; j += NT) slot[j] = -1;
    for (int j = tid; j < ni; j += NT) { ist[j] = starts[i0 + j] - p0; iln[j] = lengths[i0 + j]; }
    const int nps = np < FBBEV_NP_STAGE ? np : FBBEV_NP_STAGE;
    for (int j = tid; j < nps; j += NT) { prd[j] = rd[p0 + j]; prf[j] = rf[p0 + j]; }
    __syncthreads();
    for (int j = tid; j < ni; j += NT) {
        const int v = interval_rank[i0 + j] - v0;
        if (v >= 0 && v < nv) slot[v] = j;
    }
    __syncthreads();

    const int lpi = C / CPL;
    const int gpb = NT / lpi;
    const int g = tid / lpi, lane_slot = tid - g * lpi;
    if (g < gpb) {
        const float* fbase = feat + lane_slot * CPL;
        for (int v = g; v < nv; v += gpb) {
            const int s = slot[v];
            float acc[CPL];
            if (s >= 0) {
                fbbev_interval_sum_staged<CPL, 4>(C, ist[s], iln[s], p0, prd, prf, depth, fbase, rd, rf, acc);
            } else {
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
            }
            float* dst = obase + (long long)v * C + lane_slot * CPL;
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                fbbev_v4f val; val[0] = acc[4 * q]; val[1] = acc[4 * q + 1]; val[2] = acc[4 * q + 2]; val[3] = acc[4 * q + 3];
                fbbev_store4<ST>(dst + 4 * q, val);
            }
        }
    }
}

// ================================================================ channels-last, small tiles ("one store per thread")
// Measured on MI355X (tools/micro/fill_bench.hip, profiles/r01_fill_bench*.jsonl): the HBM write stream peaks
// (7.0-7.6 TB/s) when every workgroup writes ~4 KiB with ONE 16-byte `sc1 nt` store per thread; 4+ stores per
// thread or persistent grid-stride loops lose 15-80 %.  With the (B,Z,Y,X,C) layout a tile of TV voxels is one
// contiguous TV*C*4-byte span, so TV=16, C=80 gives 5 KiB per workgroup of 320 threads: thread -> (voxel, channel
// quad), exactly one store.  Non-empty voxels run the same in-order fmaf chain first (C/4 lanes per voxel).
//
// Tile table built by scatter instead of a binary search per tile (there are B*Z*Y*X/TV tiles):
// tile_first[T] / tile_last[T] = first / last interval whose voxel falls in tile T, -1 for empty tiles.
__global__ void __launch_bounds__(256)
k_tile_scatter(const int* __restrict__ interval_rank, const int* __restrict__ counts, int n_intervals_max,
               int tile_shift, long long n_voxels, int* __restrict__ tile_first, int* __restrict__ tile_last) {
    int n = counts[1];
    if (n > n_intervals_max) n = n_intervals_max;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = interval_rank[i];
        if (r < 0 || r >= n_voxels) continue;          // fp32-rank overflow garbage is never pooled
        const int T = r >> tile_shift;
        const int rp = (i > 0) ? interval_rank[i - 1] : -1;
        const int rn = (i + 1 < n) ? interval_rank[i + 1] : -1;
        if (rp < 0 || (rp >> tile_shift) != T) tile_first[T] = i;
        if (rn < 0 || rn >= n_voxels || (rn >> tile_shift) != T) tile_last[T] = i;
    }
}

template <int ST>
__global__ void __launch_bounds__(1024)
k_pool_fwd_cl_small(int C, int tile_shift, long long n_voxels, int n_tiles, int swizzle,
                    const float* __restrict__ depth, const float* __restrict__ feat,
                    const int* __restrict__ rd, const int* __restrict__ rf,
                    const int* __restrict__ interval_rank, const int* __restrict__ starts,
                    const int* __restrict__ lengths, const int* __restrict__ tile_first,
                    const int* __restrict__ tile_last, float* __restrict__ out) {
    // Waves are independent: no LDS, no workgroup barrier.  A wave whose voxels are all empty issues its
    // single store and retires at once, so only waves that really gather hold a wave slot.
    const int TV = 1 << tile_shift;
    const int q4 = C >> 2;                            // channel quads per voxel row
    const int tid = threadIdx.x, lane = tid & 63;
    int t = blockIdx.x;
    if (swizzle) {
        const int sh = swizzle - 1;
        const int xcd = t & 7, j = t >> 3;
        t = ((((j >> sh) << 3) + xcd) << sh) + (j & ((1 << sh) - 1));
    }
    if (t >= n_tiles) return;
    const long long v0 = (long long)t << tile_shift;
    const int nv = (n_voxels - v0 < TV) ? (int)(n_voxels - v0) : TV;
    const int v = tid / q4, quad = tid - v * q4;
    const bool active = v < nv;
    const int i0 = tile_first[t];
    float* dst = out + (v0 + v) * C + quad * 4;
    fbbev_v4f val; val[0] = val[1] = val[2] = val[3] = 0.f;
    if (i0 < 0) {                                     // empty tile: the store is all there is
        if (active) fbbev_store4<ST>(dst, val);
        return;
    }
    const int ni = tile_last[t] - i0 + 1;             // <= TV <= 64 intervals, one per lane
    const int myrank = (lane < ni) ? (interval_rank[i0 + lane] - (int)v0) : -1;
    int s = -1;
    for (int k = 0; k < ni; ++k) {                    // wave-uniform trip count
        const int r = __shfl(myrank, k, 64);
        if (r == v) s = i0 + k;
    }
    if (active && s >= 0) {
        float acc[4];
        fbbev_interval_sum<4>(C, starts[s], lengths[s], depth, feat + quad * 4, rd, rf, acc);
        val[0] = acc[0]; val[1] = acc[1]; val[2] = acc[2]; val[3] = acc[3];
    }
    if (active) fbbev_store4<ST>(dst, val);
}
