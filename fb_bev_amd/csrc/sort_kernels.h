// sort_kernels.h -- stable LSD radix sort of (key, value) pairs for the voxel ranking.
//
// The reference sorts with torch.argsort (view_transformer.py:590, unstable).  Here a hand-written
// stable radix sort on exactly the key bits that can be set (log2(B*X*Y*Z)+1), RB bits per pass:
// Pass 1 also COMPACTS: keys equal to `drop_key` (the out-of-grid sentinel, ~55 % of a frustum) are
// neither counted nor written, and the number of kept keys P is published on the device; later passes
// read their length from that device counter (no host sync) and move only P pairs.
//   k_sort_hist    : per-workgroup digit histogram (LDS atomics) -> hist[digit][workgroup]
//   k_sort_rowsum  : totals[digit] = row sum (no global atomics, no memset)
//   k_sort_scan    : one workgroup per digit: base = sum of lower digits' totals, then an exclusive scan of
//                    the digit's row over workgroups (wave-prefix-sum block scan)
//   k_sort_scatter : each wave owns a contiguous 64*ROUNDS-key chunk; per round of 64 keys the lanes holding the
//                    same digit find each other with RB ballots (a wave-level match), rank = popcount of the
//                    lower matching lanes + the wave's running digit counter (LDS, wave-private, no
//                    workgroup barrier inside the loop); after one barrier the per-digit wave prefix is
//                    added and the pairs go to their final positions.  Order = (workgroup, wave, round, lane)
//                    = input order, so the sort is stable and the voxel order canonical.
#pragma once
#include "rt.h"
#include "geom_kernels.h"
#include "rank_kernels.h"

#define FBBEV_SORT_WAVES 4
#ifndef FBBEV_SORT_ROUNDS
#define FBBEV_SORT_ROUNDS 8    // keys per lane; tile = 4 waves x 64 lanes x ROUNDS (8: 18 % faster at B=1, equal at B=16)
#endif
#define FBBEV_SORT_TILE (FBBEV_SORT_WAVES * 64 * FBBEV_SORT_ROUNDS)   // 2048 keys per workgroup
#define FBBEV_SORT_MAX_RB 9

template <int RB>
__global__ void __launch_bounds__(256)
k_sort_hist(const unsigned int* __restrict__ keys, long long n_host, const int* __restrict__ n_dev, int shift,
            int nblocks, unsigned int drop_key, int drop, int* __restrict__ hist) {
    constexpr int NB = 1 << RB;
    const long long n = n_dev ? (long long)*n_dev : n_host;
    __shared__ int cnt[NB];
    for (int d = threadIdx.x; d < NB; d += 256) cnt[d] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * FBBEV_SORT_TILE;
    constexpr int PER = FBBEV_SORT_TILE / 256;
    unsigned int key[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {                 // all loads first: one memory round trip, not PER of them
        const long long idx = base + threadIdx.x + r * 256;
        key[r] = (idx < n) ? keys[idx] : drop_key;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const long long idx = base + threadIdx.x + r * 256;
        if (idx < n && !(drop && key[r] == drop_key)) atomicAdd(&cnt[(key[r] >> shift) & (NB - 1)], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += 256) {
        const int c = cnt[d];
        hist[(long long)d * nblocks + blockIdx.x] = c;
    }
}

// grid = number of digits: totals[d] = number of keys with digit d (row sum of the histogram matrix)
__global__ void __launch_bounds__(256)
k_sort_rowsum(const int* __restrict__ hist, int nblocks, int* __restrict__ totals) {
    __shared__ int lds4[4];
    const int* row = hist + (long long)blockIdx.x * nblocks;
    int part = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) part += row[i];
    int tot;
    (void)fbbev_block_excl_scan(part, lds4, &tot);
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

// grid = number of digits; exclusive scan of hist[d][0..nblocks) offset by the totals of all lower digits
__global__ void __launch_bounds__(256)
k_sort_scan(int* __restrict__ hist, const int* __restrict__ totals, int nblocks, int* __restrict__ n_out) {
    __shared__ int lds4[4];
    const int d = blockIdx.x;
    const int NBs = (int)gridDim.x;             // digits
    if (n_out && d == NBs - 1) {                // publish the number of keys this pass keeps
        int all = 0;
        for (int j = threadIdx.x; j < NBs; j += 256) all += totals[j];
        int tot;
        (void)fbbev_block_excl_scan(all, lds4, &tot);
        if (threadIdx.x == 0) *n_out = tot;
    }
    int part = 0;
    for (int j = threadIdx.x; j < d; j += 256) part += totals[j];
    int base;
    (void)fbbev_block_excl_scan(part, lds4, &base);
    int running = base;
    int* row = hist + (long long)d * nblocks;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = (i < nblocks) ? row[i] : 0;
        int total;
        const int ex = fbbev_block_excl_scan(v, lds4, &total);
        if (i < nblocks) row[i] = running + ex;
        running += total;
    }
}

template <int RB>
__global__ void __launch_bounds__(256)
k_sort_scatter(const unsigned int* __restrict__ keys_in, const unsigned int* __restrict__ vals_in,
               long long n_host, const int* __restrict__ n_dev, long long seg_len, int chunks_per_seg, int shift,
               int nblocks, unsigned int drop_key, int drop, const int* __restrict__ hist,
               unsigned int* __restrict__ keys_out, unsigned int* __restrict__ vals_out) {
    // Tiling: workgroup = (segment, 4096-key chunk of that segment).  Flat sorts use one segment of length n;
    // the geometry-fused first pass uses one segment per camera frustum so that its histogram kernel
    // (k_sort_hist_geom) needs a single camera's matrices.  vals_in == nullptr: value = key position.
    constexpr int NB = 1 << RB;
    const long long n_all = n_dev ? (long long)*n_dev : n_host;
    const long long seg = blockIdx.x / chunks_per_seg;
    const long long seg_end = (seg + 1) * seg_len;
    const long long n = seg_end < n_all ? seg_end : n_all;
    __shared__ int cnt[FBBEV_SORT_WAVES][NB];    // per-wave running digit counters -> per-wave totals
    __shared__ int woff[FBBEV_SORT_WAVES][NB];   // global position of each wave's first key of a digit
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < FBBEV_SORT_WAVES * NB; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const long long chunk = seg * seg_len + (long long)(blockIdx.x - seg * chunks_per_seg) * FBBEV_SORT_TILE +
                            (long long)wave * (64 * FBBEV_SORT_ROUNDS);
    unsigned int k[FBBEV_SORT_ROUNDS], v[FBBEV_SORT_ROUNDS];
    int lr[FBBEV_SORT_ROUNDS];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < FBBEV_SORT_ROUNDS; ++r) {     // all loads of the wave's chunk first (one memory round trip)
        const long long idx = chunk + r * 64 + lane;
        const bool valid = idx < n;
        k[r] = valid ? keys_in[idx] : 0u;
        v[r] = valid ? (vals_in ? vals_in[idx] : (unsigned int)idx) : 0u;
    }
#pragma unroll
    for (int r = 0; r < FBBEV_SORT_ROUNDS; ++r) {
        const long long idx = chunk + r * 64 + lane;
        bool valid = idx < n;
        valid = valid && !(drop && k[r] == drop_key);
        lr[r] = -1;
        const unsigned int d = (k[r] >> shift) & (NB - 1);
        unsigned long long m = __ballot(valid ? 1 : 0);          // wave-level match on the digit
#pragma unroll
        for (int bit = 0; bit < RB; ++bit) {
            const unsigned long long b = __ballot((int)((d >> bit) & 1u));
            m &= ((d >> bit) & 1u) ? b : ~b;
        }
        const int leader = valid ? (__ffsll((long long)m) - 1) : lane;
        int prev = 0;
        if (valid && lane == leader) {
            prev = cnt[wave][d];
            cnt[wave][d] = prev + __popcll(m);
        }
        prev = __shfl(prev, leader, 64);
        if (valid) lr[r] = prev + __popcll(m & lt);
    }
    __syncthreads();
    for (int d = tid; d < NB; d += 256) {
        int run = hist[(long long)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < FBBEV_SORT_WAVES; ++w) { woff[w][d] = run; run += cnt[w][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FBBEV_SORT_ROUNDS; ++r) {
        if (lr[r] >= 0) {
            const unsigned int d = (k[r] >> shift) & (NB - 1);
            const int pos = woff[wave][d] + lr[r];
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
    }
}

// ---------------------------------------------------------------- pass 1 with a fused geometry -> key source
// k_sort_hist_geom: workgroup = (camera, 4096-point chunk of that camera's D*H*W frustum).  The keys are
// evaluated in registers from the camera parameters (fbbev_point_coor + fbbev_rank_key: the same code the
// two-step contract path runs) -- `coor` and the point-id array are never materialised -- histogrammed, and
// stored (4 B/point) for the scatter, which runs with the same per-camera tiling and implicit point ids.
struct fbbev_geom_src {
    fbbev_cam_ptrs cam;
    const float* frustum;    // optional (D,H,W,3) template (u, v, depth) of create_frustum: table lookup instead of
                             // three runtime integer divisions per point
    fbbev_grid_params gp;
    unsigned int sentinel;
    int chunks_per_cam;      // ceil(D*H*W / FBBEV_SORT_TILE)
};

__device__ __forceinline__ unsigned int fbbev_geom_key(const fbbev_geom_src& g, const float* m, int cam, int i) {
    float u, v, dep;
    if (g.frustum) {
        u = g.frustum[3 * i]; v = g.frustum[3 * i + 1]; dep = g.frustum[3 * i + 2];
    } else {
        const int w = i % g.cam.W, h = (i / g.cam.W) % g.cam.H, d = i / (g.cam.W * g.cam.H);
        u = g.cam.xs[w]; v = g.cam.ys[h]; dep = g.cam.ds[d];
    }
    float cx, cy, cz;
    fbbev_point_coor(m, u, v, dep, cx, cy, cz);
    return fbbev_rank_key(cx, cy, cz, g.gp, (float)(cam / g.cam.N), g.sentinel);
}

template <int RB>
__global__ void __launch_bounds__(256)
k_sort_hist_geom(fbbev_geom_src g, int shift, int nblocks, int* __restrict__ hist,
                 unsigned int* __restrict__ keys_out) {
    constexpr int NB = 1 << RB;
    __shared__ int cnt[NB];
    __shared__ float m[33];
    const int cam = blockIdx.x / g.chunks_per_cam, chunk = blockIdx.x - cam * g.chunks_per_cam;
    if (threadIdx.x == 0) fbbev_cam_setup(g.cam, cam, m);
    for (int d = threadIdx.x; d < NB; d += 256) cnt[d] = 0;
    __syncthreads();
    const int dhw = g.cam.D * g.cam.H * g.cam.W;
    const int base = chunk * FBBEV_SORT_TILE;
    constexpr int PER = FBBEV_SORT_TILE / 256;
    unsigned int key[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {                 // keys of all PER points first (their table loads overlap) ...
        const int idx = base + threadIdx.x + r * 256;
        key[r] = (idx < dhw) ? fbbev_geom_key(g, m, cam, idx) : g.sentinel;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {                 // ... then the stores and the LDS histogram
        const int idx = base + threadIdx.x + r * 256;
        if (idx < dhw) {
            keys_out[(long long)cam * dhw + idx] = key[r];
            if (key[r] != g.sentinel) atomicAdd(&cnt[(key[r] >> shift) & (NB - 1)], 1);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += 256) {
        const int c = cnt[d];
        hist[(long long)d * nblocks + blockIdx.x] = c;
    }
}
