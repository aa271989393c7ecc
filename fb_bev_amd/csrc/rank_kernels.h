// rank_kernels.h -- voxel ranking: frustum point -> voxel key -> sorted runs (intervals).
//
// Replaces LSSViewTransformerFunction3D.voxel_pooling_prepare_v2
// (fbbev/view_transformation/forward_projection/view_transformer.py:547-605): in the reference
// ~17 torch launches, an unstable argsort over B*N*D*H*W keys, three boolean-mask gathers and a
// torch.where -- at least four host syncs.  Here everything stays on the device:
//   k_keys_hist_*     : point -> key (fp32 rank evaluation + truncation, bit-for-bit the
//                        reference arithmetic); points outside the grid get a sentinel key  (sort_kernels.h)
//   (stable radix sort of (key, point id) pairs, sort_kernels.h: count matrix + scatter per pass; its first
//    pass drops the sentinel keys and publishes P, the number of kept points, on the device)
//   k_interval_count / k_interval_write : run heads of the sorted keys -> interval_starts / interval_lengths /
//                        interval_rank, ranks_feat, I (number of intervals); see the kernels below
#pragma once
#include "rt.h"


struct fbbev_grid_params {
    float lx, ly, lz;     // grid_lower_bound        (view_transformer.py:384)
    float ix, iy, iz;     // grid_interval           (:385)
    float gx, gy, gz;     // grid_size as fp32       (:386-387)
    float f_yx, f_zyx;    // fl(gy*gx), fl(fl(gz*gy)*gx): the 0-dim fp32 products of :586-588
};

// view_transformer.py:570-589.  Every operation is a single correctly-rounded fp32 op (no fma
// contraction, IEEE divide): the rank must be the SAME float the reference computes, because it
// is the sort key and, above 2^24, decides which voxels collide.
__device__ __forceinline__ unsigned int fbbev_rank_key(float cx, float cy, float cz, const fbbev_grid_params& gp,
                                                       float bf, unsigned int sentinel) {
    const float fx = __fdiv_rn(fbbev_sub(cx, gp.lx), gp.ix);
    const float fy = __fdiv_rn(fbbev_sub(cy, gp.ly), gp.iy);
    const float fz = __fdiv_rn(fbbev_sub(cz, gp.lz), gp.iz);
    // .long(): truncation toward zero.  |f| >= 2^31 or NaN can never be inside the grid.
    const bool finite = (fx == fx) && (fy == fy) && (fz == fz) && fabsf(fx) < 2.0e9f &&
                        fabsf(fy) < 2.0e9f && fabsf(fz) < 2.0e9f;
    const int vx = finite ? (int)fx : -1;
    const int vy = finite ? (int)fy : -1;
    const int vz = finite ? (int)fz : -1;
    // kept: integer >= 0 and (float)v < grid_size (long vs 0-dim fp32 tensor compares in fp32)
    const bool kept = finite && vx >= 0 && vy >= 0 && vz >= 0 && (float)vx < gp.gx &&
                      (float)vy < gp.gy && (float)vz < gp.gz;
    float r = fbbev_mul(bf, gp.f_zyx);
    r = fbbev_add(r, fbbev_mul((float)vz, gp.f_yx));
    const float t = fbbev_add(fbbev_mul((float)vy, gp.gx), (float)vx);
    r = fbbev_add(r, t);
    return kept ? (unsigned int)(int)r : sentinel;
}

__device__ __forceinline__ int fbbev_wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// exclusive scan of one int per thread over a 256-thread block; *total gets the block sum.
// lds4: 4 ints of static LDS owned by the caller.
__device__ __forceinline__ int fbbev_block_excl_scan(int v, int* lds4, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int inc = fbbev_wave_incl_scan(v, lane);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = lds4[i];
        if (i < w) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// exclusive sum / max scans of one int per thread over a block of WAVES wave64s; *total = block result.
// ldsw: WAVES ints of LDS owned by the caller.  (max: values >= 0, identity 0)
template <int WAVES, bool MAX>
__device__ __forceinline__ int fbbev_block_excl_scan_w(int v, int* ldsw, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int nbr = __shfl_up(inc, o, 64);
        if (lane >= o) inc = MAX ? (nbr > inc ? nbr : inc) : inc + nbr;
    }
    int excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = 0;
    if (lane == 63) ldsw[w] = inc;
    __syncthreads();
    int carry = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        const int x = ldsw[i];
        if (i < w) carry = MAX ? (x > carry ? x : carry) : carry + x;
        tot = MAX ? (x > tot ? x : tot) : tot + x;
    }
    __syncthreads();
    *total = tot;
    return MAX ? (excl > carry ? excl : carry) : carry + excl;
}

// ---------------------------------------------------------------- run detection: two launches, no inter-workgroup waiting
// keys[0..P) sorted ascending, vals = point ids; P = counts[0] (published by the sort's first pass).  Chunk w of both
// kernels = keys[w*TILE, (w+1)*TILE), TILE = WAVES*64*ITEMS, thread t owns ITEMS consecutive keys.
//   k_interval_count : per chunk (number of run heads, position + 1 of its last head or 0) -> chunk_info[w]
//   k_interval_write : the interval index of a head is a prefix sum over the whole array and its length needs the position of
//                      the head before it: both come from a plain read of chunk_info[0..w) (a few hundred pairs, one
//                      coalesced load + two block reductions) -- no scan launch, no look-back chain (measured on MI355X:
//                      a decoupled look-back over device-scope status words cost 33 us here, profiles/r02_*).
//                      Writes interval_starts / interval_lengths / interval_rank, ranks_feat and I.
// ITEMS consecutive 32-bit words starting at p[base] into r[]: 16-byte loads when the whole run is inside [0, P) and
// the address is 16-byte aligned (a thread owning 8 consecutive keys otherwise issues 8 dword loads whose 64 lanes are
// 32 bytes apart -- every instruction touches 16 cache lines for 256 useful bytes); words at or beyond P read as `fill`.
template <int ITEMS>
__device__ __forceinline__ void fbbev_ld_run(const unsigned int* __restrict__ p, long long base, long long P, unsigned int fill,
                                             unsigned int (&r)[ITEMS]) {
    static_assert(ITEMS % 4 == 0, "runs of whole 16-byte groups");
    if (base + ITEMS <= P && ((reinterpret_cast<uintptr_t>(p + base) & 15u) == 0)) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; ++q) {
            const fbbev_v4u t = *reinterpret_cast<const fbbev_v4u*>(p + base + 4 * q);
            r[4 * q] = t[0]; r[4 * q + 1] = t[1]; r[4 * q + 2] = t[2]; r[4 * q + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) r[j] = (base + j < P) ? p[base + j] : fill;
    }
}

template <int WAVES, int ITEMS>
__global__ void __launch_bounds__(WAVES * 64)
k_interval_count(const unsigned int* __restrict__ keys, const int* __restrict__ counts, const int* __restrict__ skip,
                 int2* __restrict__ chunk_info) {
    if (skip != nullptr && *skip != 0) return;
    constexpr int NT = WAVES * 64;
    constexpr int TILE = NT * ITEMS;
    __shared__ int ldsw[WAVES];
    const int tid = threadIdx.x;
    const int P = counts[0];
    const long long wg0 = (long long)blockIdx.x * TILE;
    if (wg0 >= P) return;
    const long long base = wg0 + (long long)tid * ITEMS;
    int local = 0, last1 = 0;
    unsigned int prevk = (base > 0 && base <= P) ? keys[base - 1] : 0u;
    unsigned int kk[ITEMS];
    fbbev_ld_run<ITEMS>(keys, base, P, 0u, kk);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const long long i = base + j;
        if (i < P) {
            const unsigned int k = kk[j];
            if (i == 0 || prevk != k) { ++local; last1 = (int)i + 1; }
            prevk = k;
        }
    }
    int tot, wg_last1;
    (void)fbbev_block_excl_scan_w<WAVES, false>(local, ldsw, &tot);
    (void)fbbev_block_excl_scan_w<WAVES, true>(last1, ldsw, &wg_last1);
    if (tid == 0) chunk_info[blockIdx.x] = make_int2(tot, wg_last1);
}

// division of a 32-bit unsigned by an invariant divisor d (1 <= d < 2^31): q = (t + ((n - t) >> 1)) >> sh with
// t = mulhi(n, m) -- Granlund & Montgomery; the launcher computes (m, sh) once.  Replaces two hardware-less integer
// divisions per point (~40 dependent instructions each, a third of k_interval_write's issue stalls in the round-2 PMC run).
struct fbbev_fastdiv { unsigned int d, m, sh; };
__device__ __forceinline__ unsigned int fbbev_div(unsigned int n, const fbbev_fastdiv& f) {
    if (f.d == 1u) return n;
    const unsigned int t = (unsigned int)(((unsigned long long)n * f.m) >> 32);
    return (t + ((n - t) >> 1)) >> f.sh;
}

template <int WAVES, int ITEMS>
__global__ void __launch_bounds__(WAVES * 64)
k_interval_write(const unsigned int* __restrict__ keys, const unsigned int* __restrict__ vals, fbbev_fastdiv div_dhw,
                 fbbev_fastdiv div_hw, const int2* __restrict__ chunk_info, const int* __restrict__ skip, int* __restrict__ ranks_feat,
                 int* __restrict__ interval_starts, int* __restrict__ interval_lengths, int* __restrict__ interval_rank,
                 int* __restrict__ counts) {
    if (skip != nullptr && *skip != 0) return;
    constexpr int NT = WAVES * 64;
    constexpr int TILE = NT * ITEMS;
    __shared__ int ldsw[WAVES];
    const int tid = threadIdx.x;
    const int wg = blockIdx.x;
    const int P = counts[0];
    const long long wg0 = (long long)wg * TILE;
    if (wg0 >= P) {
        if (wg == 0 && tid == 0) counts[1] = 0;          // no point inside the grid: P = I = 0
        return;
    }
    // heads / last head of all earlier chunks: one strided pass over chunk_info[0..wg) + two block reductions
    int h = 0, l1 = 0;
    for (int r = tid; r < wg; r += NT) {
        const int2 ci = chunk_info[r];
        h += ci.x;
        l1 = ci.y > l1 ? ci.y : l1;                       // positions grow with the chunk index: max = nearest
    }
    int before_heads, before_last1;
    (void)fbbev_block_excl_scan_w<WAVES, false>(h, ldsw, &before_heads);
    (void)fbbev_block_excl_scan_w<WAVES, true>(l1, ldsw, &before_last1);
    const long long base = wg0 + (long long)tid * ITEMS;
    bool head[ITEMS];
    unsigned int key[ITEMS];
    int local = 0, last1 = 0;
    unsigned int prevk = (base > 0 && base <= P) ? keys[base - 1] : 0u;
    unsigned int pid[ITEMS];
    fbbev_ld_run<ITEMS>(keys, base, P, 0u, key);
    fbbev_ld_run<ITEMS>(vals, base, P, 0u, pid);
    int rf[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const long long i = base + j;
        head[j] = false;
        rf[j] = 0;
        if (i < P) {
            const unsigned int k = key[j];
            head[j] = (i == 0 || prevk != k);
            prevk = k;
            if (head[j]) { ++local; last1 = (int)i + 1; }
            // view_transformer.py:563-568: feature pixel of point ((b*N+n)*D+d)*HW + hw
            const unsigned int cam = fbbev_div(pid[j], div_dhw), hw = pid[j] - fbbev_div(pid[j], div_hw) * div_hw.d;
            rf[j] = (int)(cam * div_hw.d + hw);
        }
    }
    if (base + ITEMS <= P && ((reinterpret_cast<uintptr_t>(ranks_feat + base) & 15u) == 0)) {
#pragma unroll
        for (int q = 0; q < ITEMS / 4; ++q) {
            fbbev_v4i t;
            t[0] = rf[4 * q]; t[1] = rf[4 * q + 1]; t[2] = rf[4 * q + 2]; t[3] = rf[4 * q + 3];
            *reinterpret_cast<fbbev_v4i*>(ranks_feat + base + 4 * q) = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (base + j < P) ranks_feat[base + j] = rf[j];
    }
    int tot, wg_last1;
    const int off = fbbev_block_excl_scan_w<WAVES, false>(local, ldsw, &tot);
    (void)fbbev_block_excl_scan_w<WAVES, true>(last1, ldsw, &wg_last1);
    // the chunk's heads, compacted in LDS (position, voxel), then written with consecutive lanes on consecutive interval
    // slots: a thread storing its own ~5 heads one by one scatters every store instruction over 64 x 20-byte strides
    int* hpos = reinterpret_cast<int*>(fbbev_dyn_lds_f32());        // [TILE] positions, then [TILE] voxel ranks
    int* hkey = hpos + TILE;
    {
        int o = off;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (head[j]) { hpos[o] = (int)(base + j); hkey[o] = (int)key[j]; ++o; }
    }
    __syncthreads();
    for (int o = tid; o < tot; o += NT) {
        const int pos = hpos[o], j0 = before_heads + o;
        interval_starts[j0] = pos;
        if (interval_rank) interval_rank[j0] = hkey[o];
        if (j0 > 0) interval_lengths[j0 - 1] = pos - (o > 0 ? hpos[o - 1] : before_last1 - 1);
    }
    if (tid == 0 && wg0 + TILE >= P) {                    // the chunk that holds the last point closes the chain
        const int I = before_heads + tot;
        const int lasthead1 = wg_last1 ? wg_last1 : before_last1;
        interval_lengths[I - 1] = P - (lasthead1 - 1);
        counts[1] = I;
    }
}
