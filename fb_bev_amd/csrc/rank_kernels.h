// rank_kernels.h -- voxel ranking: frustum point -> voxel key -> sorted runs (intervals).
//
// Replaces LSSViewTransformerFunction3D.voxel_pooling_prepare_v2
// (fbbev/view_transformation/forward_projection/view_transformer.py:547-605): in the reference
// ~17 torch launches, an unstable argsort over B*N*D*H*W keys, three boolean-mask gathers and a
// torch.where -- at least four host syncs.  Here everything stays on the device:
//   k_rank_keys        : point -> key (fp32 rank evaluation + truncation, bit-for-bit the
//                        reference arithmetic); points outside the grid get a sentinel key.
//   (stable radix sort of (key, point id) pairs, sort_kernels.h; its first pass drops the
//    sentinel keys and publishes P, the number of kept points, on the device)
//   k_flag_count       : per-block count of run heads
//   k_write_intervals  : prefix of the block counts + run heads -> interval_starts (wave prefix sums),
//                        ranks_feat, I (number of intervals)
//   k_interval_lengths : starts -> lengths
#pragma once
#include "rt.h"

#define FBBEV_RANK_ITEMS 4          // items per thread in the run-detection kernels
#define FBBEV_RANK_BLOCK 256
#define FBBEV_RANK_CHUNK (FBBEV_RANK_ITEMS * FBBEV_RANK_BLOCK)

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
    const float fx = __fdiv_rn(__fsub_rn(cx, gp.lx), gp.ix);
    const float fy = __fdiv_rn(__fsub_rn(cy, gp.ly), gp.iy);
    const float fz = __fdiv_rn(__fsub_rn(cz, gp.lz), gp.iz);
    // .long(): truncation toward zero.  |f| >= 2^31 or NaN can never be inside the grid.
    const bool finite = (fx == fx) && (fy == fy) && (fz == fz) && fabsf(fx) < 2.0e9f &&
                        fabsf(fy) < 2.0e9f && fabsf(fz) < 2.0e9f;
    const int vx = finite ? (int)fx : -1;
    const int vy = finite ? (int)fy : -1;
    const int vz = finite ? (int)fz : -1;
    // kept: integer >= 0 and (float)v < grid_size (long vs 0-dim fp32 tensor compares in fp32)
    const bool kept = finite && vx >= 0 && vy >= 0 && vz >= 0 && (float)vx < gp.gx &&
                      (float)vy < gp.gy && (float)vz < gp.gz;
    float r = __fmul_rn(bf, gp.f_zyx);
    r = __fadd_rn(r, __fmul_rn((float)vz, gp.f_yx));
    const float t = __fadd_rn(__fmul_rn((float)vy, gp.gx), (float)vx);
    r = __fadd_rn(r, t);
    return kept ? (unsigned int)(int)r : sentinel;
}

__global__ void __launch_bounds__(256)
k_rank_keys(const float* __restrict__ coor, long long npts, long long pts_per_batch,
            fbbev_grid_params gp, unsigned int sentinel, const float* __restrict__ depth, float depth_thr,
            unsigned int* __restrict__ keys, unsigned int* __restrict__ vals) {
    for (long long pid = (long long)blockIdx.x * blockDim.x + threadIdx.x; pid < npts;
         pid += (long long)gridDim.x * blockDim.x) {
        unsigned int key = fbbev_rank_key(coor[3 * pid], coor[3 * pid + 1], coor[3 * pid + 2], gp,
                                          (float)(pid / pts_per_batch), sentinel);
        // BEVDet-era variant (mmdet3d/models/necks/view_transformer.py:556-557): kept &= depth.view(-1) > 0.01 --
        // the number of kept points becomes data dependent, which the device-side counts absorb
        if (depth && !(depth[pid] > depth_thr)) key = sentinel;
        keys[pid] = key;
        vals[pid] = (unsigned int)pid;
    }
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

// keys[0..P) sorted ascending (P = counts[0], published by the sort's compaction pass)
__global__ void __launch_bounds__(FBBEV_RANK_BLOCK)
k_flag_count(const unsigned int* __restrict__ keys, const int* __restrict__ counts,
             int* __restrict__ block_counts) {
    __shared__ int lds4[4];
    const long long n = counts[0];
    const long long base = (long long)blockIdx.x * FBBEV_RANK_CHUNK + threadIdx.x * FBBEV_RANK_ITEMS;
    int local = 0;
#pragma unroll
    for (int j = 0; j < FBBEV_RANK_ITEMS; ++j) {
        const long long i = base + j;
        if (i < n) {
            const unsigned int k = keys[i];
            local += (i == 0 || keys[i - 1] != k) ? 1 : 0;
        }
    }
    int total;
    (void)fbbev_block_excl_scan(local, lds4, &total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(FBBEV_RANK_BLOCK)
k_write_intervals(const unsigned int* __restrict__ keys, const unsigned int* __restrict__ vals,
                  int* __restrict__ counts, const int* __restrict__ block_counts, int nblocks,
                  int D, int HW, int* __restrict__ ranks_feat, int* __restrict__ interval_starts,
                  int* __restrict__ interval_rank) {
    __shared__ int lds4[4];
    const long long n = counts[0];
    // exclusive prefix of the per-block head counts: every block sums the counts of the blocks before it
    // (<= 16 KiB of L2-resident ints) -- cheaper than a separate single-block scan launch on the critical path
    int part = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += FBBEV_RANK_BLOCK) part += block_counts[j];
    int block_offset;
    (void)fbbev_block_excl_scan(part, lds4, &block_offset);
    const long long base = (long long)blockIdx.x * FBBEV_RANK_CHUNK + threadIdx.x * FBBEV_RANK_ITEMS;
    bool head[FBBEV_RANK_ITEMS];
    unsigned int key[FBBEV_RANK_ITEMS];
    int local = 0;
    const unsigned int dhw = (unsigned int)D * (unsigned int)HW;
#pragma unroll
    for (int j = 0; j < FBBEV_RANK_ITEMS; ++j) {
        const long long i = base + j;
        head[j] = false;
        key[j] = 0u;
        if (i < n) {
            const unsigned int k = keys[i];
            key[j] = k;
            head[j] = (i == 0 || keys[i - 1] != k);
            local += head[j] ? 1 : 0;
            // view_transformer.py:563-568: feature pixel of point ((b*N+n)*D+d)*HW + hw
            const unsigned int pid = vals[i];
            ranks_feat[i] = (int)((pid / dhw) * (unsigned int)HW + pid % (unsigned int)HW);
        }
    }
    int total;
    int j0 = block_offset + fbbev_block_excl_scan(local, lds4, &total);
    if (blockIdx.x == (unsigned)(nblocks - 1) && threadIdx.x == 0) counts[1] = block_offset + total;   // I
#pragma unroll
    for (int j = 0; j < FBBEV_RANK_ITEMS; ++j) {
        if (head[j]) {
            interval_starts[j0] = (int)(base + j);
            if (interval_rank) interval_rank[j0] = (int)key[j];
            ++j0;
        }
    }
}

__global__ void __launch_bounds__(256)
k_interval_lengths(const int* __restrict__ interval_starts, const int* __restrict__ counts,
                   long long n_max, int* __restrict__ interval_lengths) {
    const int P = counts[0], I = counts[1];
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < I && j < n_max;
         j += (long long)gridDim.x * blockDim.x) {
        const int s = interval_starts[j];
        const int e = (j + 1 < I) ? interval_starts[j + 1] : P;
        interval_lengths[j] = e - s;
    }
}
