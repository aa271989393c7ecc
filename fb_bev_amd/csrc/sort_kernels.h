// sort_kernels.h -- stable LSD radix sort of (voxel key, point id) pairs for the voxel ranking.
//
// The reference sorts with torch.argsort (view_transformer.py:590, unstable).  Here a hand-written stable radix sort on
// exactly the key bits that can be set (log2(B*X*Y*Z)), <= 8 bits per pass, built for launch count and latency -- the
// whole ranking chain is ~150 MB of traffic, so it is bound by the number of dependent launches, not by HBM.  Two
// launches per pass, no scan kernels, no inter-workgroup waiting:
//   k_keys_hist_geom / k_keys_hist_coor : evaluate the keys (from the camera geometry, or from a materialised `coor`),
//                      store them (4 B/point) and count the pass-0 digits per CHUNK -> count matrix row [chunk][digit]
//   k_sort_hist      : the same count matrix for a later pass (reads the keys only)
//   k_sort_scatter   : a workgroup owns one contiguous chunk of WAVES*64*ROUNDS pairs.  Prologue: the exclusive prefix of
//                      its digits over the EARLIER chunks and the digit totals are plain column sums of the count matrix
//                      -- chunks are fat (up to 12288 pairs), so the matrix is a few hundred KB, L2-resident, and summing
//                      it directly is cheaper than a scan launch or a look-back chain (measured on MI355X: decoupled
//                      look-back reads its status words with device-scope loads that miss the per-XCD L2; 67 us per pass).
//                      Then it loads its pairs into registers, ranks them stably (each wave ranks its 64 keys per round with
//                      wave ballots = a wave-level match, on top of a wave-private running digit counter in LDS) and
//                      scatters.  Pass 0 also COMPACTS: out-of-grid points (~55 % of a frustum) carry the drop key, are
//                      neither counted nor written, and P is published on the device.
// Order = (chunk, wave, round, lane) = input order, so the sort is stable and the voxel order canonical.
#pragma once
#include "rt.h"
#include "geom_kernels.h"
#include "rank_kernels.h"

#define FBBEV_SORT_MAX_RB 8
#define FBBEV_SORT_MAX_NB (1 << FBBEV_SORT_MAX_RB)
#define FBBEV_SORT_MAX_PASSES 4
#define FBBEV_DROP_KEY 0xffffffffu     // out-of-grid sentinel: above every rank (ranks < 2^30)

// early-out shared by every kernel of a cached rank build: *skip != 0 <=> the camera parameters equal the cached ones
__device__ __forceinline__ bool fbbev_skip(const int* skip) { return skip != nullptr && *skip != 0; }

// ---------------------------------------------------------------- keys from the camera geometry
// Workgroup w owns the flat point range [w*T, (w+1)*T) -- the chunk the pass-0 scatter will own -- and walks the cameras
// it intersects (one or two for real frusta).  The keys are evaluated in registers from the camera parameters
// (fbbev_point_coor + fbbev_rank_key: the same code the two-step contract path runs) -- `coor` and the point-id array
// are never materialised.
struct fbbev_geom_src {
    fbbev_cam_ptrs cam;
    const float* frustum;    // optional (D,H,W,3) template (u, v, depth) of create_frustum: table lookup instead of
                             // three runtime integer divisions per point
    fbbev_grid_params gp;
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
    return fbbev_rank_key(cx, cy, cz, g.gp, (float)(cam / g.cam.N), FBBEV_DROP_KEY);
}

template <int NT, int PER>
__global__ void __launch_bounds__(NT)
k_keys_hist_geom(fbbev_geom_src g, long long n, int rb, const int* __restrict__ skip,
                 unsigned int* __restrict__ keys_out, int* __restrict__ matrix) {
    if (fbbev_skip(skip)) return;
    __shared__ int cnt[FBBEV_SORT_MAX_NB];
    __shared__ float m[33];
    const int nb = 1 << rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    for (int d = threadIdx.x; d < nb; d += NT) cnt[d] = 0;
    // 32-bit point ids: n < 2^30 (checked by the launcher)
    const int dhw = g.cam.D * g.cam.H * g.cam.W;
    const int base = blockIdx.x * (NT * PER);
    const int end = (long long)base + NT * PER > n ? (int)n : base + NT * PER;
    const int cam_lo = base / dhw, cam_hi = (end - 1) / dhw;
    constexpr int G = 4;                                // points in flight per thread (register budget of 1024-thread workgroups)
    for (int cam = cam_lo; cam <= cam_hi; ++cam) {
        __syncthreads();                                // previous camera's readers of m[] are done (and cnt[] is zeroed)
        if (threadIdx.x == 0) fbbev_cam_setup(g.cam, cam, m);
        __syncthreads();
        const int c0 = cam * dhw;
        const int lo = base > c0 ? base : c0, hi = end < c0 + dhw ? end : c0 + dhw;   // this camera's part of the chunk
        for (int r0 = 0; r0 < PER; r0 += G) {
            unsigned int key[G];
#pragma unroll
            for (int r = 0; r < G; ++r) {               // keys of G points first (their table loads overlap) ...
                const int pid = base + (int)threadIdx.x + (r0 + r) * NT;
                key[r] = (pid >= lo && pid < hi) ? fbbev_geom_key(g, m, cam, pid - c0) : FBBEV_DROP_KEY;
            }
#pragma unroll
            for (int r = 0; r < G; ++r) {               // ... then the stores and the LDS histogram
                const int pid = base + (int)threadIdx.x + (r0 + r) * NT;
                if (pid >= lo && pid < hi) {
                    keys_out[pid] = key[r];
                    if (key[r] != FBBEV_DROP_KEY) atomicAdd(&cnt[key[r] & dmask], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += NT) matrix[(long long)blockIdx.x * nb + d] = cnt[d];
}

// ---------------------------------------------------------------- keys from a materialised coor (two-step contract)
template <int NT, int PER>
__global__ void __launch_bounds__(NT)
k_keys_hist_coor(const float* __restrict__ coor, long long npts, long long pts_per_batch, fbbev_grid_params gp,
                 const float* __restrict__ depth, float depth_thr, int rb,
                 unsigned int* __restrict__ keys_out, int* __restrict__ matrix) {
    __shared__ int cnt[FBBEV_SORT_MAX_NB];
    const int nb = 1 << rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    for (int d = threadIdx.x; d < nb; d += NT) cnt[d] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * (NT * PER);
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const long long pid = base + threadIdx.x + r * NT;
        if (pid < npts) {
            unsigned int key = fbbev_rank_key(coor[3 * pid], coor[3 * pid + 1], coor[3 * pid + 2], gp,
                                              (float)(pid / pts_per_batch), FBBEV_DROP_KEY);
            // BEVDet-era variant (mmdet3d/models/necks/view_transformer.py:556-557): kept &= depth.view(-1) > 0.01 --
            // the number of kept points becomes data dependent, which the device-side counts absorb
            if (depth && !(depth[pid] > depth_thr)) key = FBBEV_DROP_KEY;
            keys_out[pid] = key;
            if (key != FBBEV_DROP_KEY) atomicAdd(&cnt[key & dmask], 1);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += NT) matrix[(long long)blockIdx.x * nb + d] = cnt[d];
}

// count matrix of a later pass: chunk w = keys[w*T, (w+1)*T) of the P = counts[0] pairs the previous pass wrote
template <int NT, int PER>
__global__ void __launch_bounds__(NT)
k_sort_hist(const unsigned int* __restrict__ keys, const int* __restrict__ counts, int shift, int rb,
            const int* __restrict__ skip, int* __restrict__ matrix) {
    if (fbbev_skip(skip)) return;
    __shared__ int cnt[FBBEV_SORT_MAX_NB];
    const int nb = 1 << rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    const long long n = counts[0];
    const long long base = (long long)blockIdx.x * (NT * PER);
    if (base >= n) return;
    for (int d = threadIdx.x; d < nb; d += NT) cnt[d] = 0;
    __syncthreads();
    unsigned int key[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {                     // all loads first: one memory round trip, not PER of them
        const long long idx = base + threadIdx.x + r * NT;
        key[r] = (idx < n) ? keys[idx] : FBBEV_DROP_KEY;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r)
        if (key[r] != FBBEV_DROP_KEY) atomicAdd(&cnt[(key[r] >> shift) & dmask], 1);
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += NT) matrix[(long long)blockIdx.x * nb + d] = cnt[d];
}

// One scatter pass.  grid = ceil(n_max / TILE) workgroups of WAVES*64 threads, TILE = WAVES*64*ROUNDS = the chunk of
// the count matrix rows.
//   keys_in / vals_in : pairs of this pass (vals_in == nullptr: value = position = point id, pass 0)
//   matrix            : [chunks][nb] digit counts of this pass's input order
//   counts            : pass 0 publishes counts[0] = P (the number of kept pairs); later passes read it
template <int WAVES, int ROUNDS>
__global__ void __launch_bounds__(WAVES * 64)
k_sort_scatter(const unsigned int* __restrict__ keys_in, const unsigned int* __restrict__ vals_in, long long n_host,
               const int* __restrict__ matrix, int pass, int rb, const int* __restrict__ skip,
               unsigned int* __restrict__ keys_out, unsigned int* __restrict__ vals_out, int* __restrict__ counts) {
    if (fbbev_skip(skip)) return;
    constexpr int NT = WAVES * 64;
    constexpr int TILE = NT * ROUNDS;
    constexpr int NBM = FBBEV_SORT_MAX_NB;
    __shared__ int cnt[WAVES][NBM];      // per-wave running digit counters -> per-wave totals
    __shared__ int woff[WAVES][NBM];     // position of each wave's first key of a digit
    __shared__ int pos0[NBM];            // global position of this chunk's first key of a digit
    __shared__ int ldsw[WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = 1 << rb, shift = pass * rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    const int wg = blockIdx.x;
    const long long n = (pass == 0) ? n_host : (long long)counts[0];
    const long long chunk0 = (long long)wg * TILE;
    if (chunk0 >= n) {
        if (pass == 0 && n_host <= 0 && wg == 0 && tid == 0) counts[0] = 0;
        return;
    }
    for (int i = tid; i < WAVES * NBM; i += NT) (&cnt[0][0])[i] = 0;
    // prologue: column sums of the count matrix.  NT / nb row groups run in parallel (row loads are coalesced, nb ints)
    const int rows = (int)((n + TILE - 1) / TILE);
    {
        const int groups = NT / nb > 0 ? NT / nb : 1;
        int pre = 0, all = 0;
        if (tid < groups * nb) {
            const int d = tid % nb, rg = tid / nb;
            constexpr int U = 8;                          // row loads in flight per thread (the loop is latency bound)
            int r = rg;
            for (; r + (U - 1) * groups < rows; r += U * groups) {
                int c[U];
#pragma unroll
                for (int u = 0; u < U; ++u) c[u] = matrix[(long long)(r + u * groups) * nb + d];
#pragma unroll
                for (int u = 0; u < U; ++u) { all += c[u]; if (r + u * groups < wg) pre += c[u]; }
            }
            for (; r < rows; r += groups) {
                const int c = matrix[(long long)r * nb + d];
                all += c;
                if (r < wg) pre += c;
            }
        }
        // reduce the row groups through LDS (woff is free until the ranking is done)
        int* red = &woff[0][0];
        __syncthreads();                                  // cnt zero-fill
        if (tid < groups * nb) { red[tid] = pre; red[groups * nb + tid] = all; }
        __syncthreads();
        int mypre = 0, myall = 0;
        if (tid < nb) {
            for (int gi = 0; gi < groups; ++gi) { mypre += red[gi * nb + tid]; myall += red[groups * nb + gi * nb + tid]; }
        }
        __syncthreads();
        int total;
        const int dbase = fbbev_block_excl_scan_w<WAVES, false>(tid < nb ? myall : 0, ldsw, &total);
        if (tid < nb) pos0[tid] = dbase + mypre;
        if (pass == 0 && wg == 0 && tid == 0) counts[0] = total;      // P
    }
    const long long chunk = chunk0 + (long long)wave * (64 * ROUNDS);
    unsigned int k[ROUNDS], v[ROUNDS];
    int lr[ROUNDS];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {                    // all loads of the wave's chunk first (one memory round trip)
        const long long idx = chunk + r * 64 + lane;
        const bool valid = idx < n;
        k[r] = valid ? keys_in[idx] : FBBEV_DROP_KEY;
        v[r] = valid ? (vals_in ? vals_in[idx] : (unsigned int)idx) : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool valid = k[r] != FBBEV_DROP_KEY;
        lr[r] = -1;
        const unsigned int d = (k[r] >> shift) & dmask;
        unsigned long long mm = __ballot(valid ? 1 : 0);  // wave-level match on the digit
#pragma unroll
        for (int bit = 0; bit < FBBEV_SORT_MAX_RB; ++bit) {
            if (bit < rb) {
                const unsigned long long b = __ballot((int)((d >> bit) & 1u));
                mm &= ((d >> bit) & 1u) ? b : ~b;
            }
        }
        const int leader = valid ? (__ffsll((long long)mm) - 1) : lane;
        int prev = 0;
        if (valid && lane == leader) {
            prev = cnt[wave][d];
            cnt[wave][d] = prev + __popcll(mm);
        }
        prev = __shfl(prev, leader, 64);
        if (valid) lr[r] = prev + __popcll(mm & lt);
    }
    __syncthreads();
    for (int d = tid; d < nb; d += NT) {
        int run = pos0[d];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { woff[w][d] = run; run += cnt[w][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (lr[r] >= 0) {
            const unsigned int d = (k[r] >> shift) & dmask;
            const int pos = woff[wave][d] + lr[r];
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
    }
}

// ---------------------------------------------------------------- camera-parameter key of the cached index set
// The index tensors depend only on the camera parameters (and on the module's static grid / frustum): SURVEY 8f-2,
// view_transformer.py:607-611 (`pre_compute`, disabled upstream because nothing invalidates it).  One workgroup compares
// the bits of the six camera tensors with the cached copy ON THE DEVICE: state[0] = 1 (skip: every kernel of the
// build returns at once, the previous index set stays valid) or 0 (key refreshed, build runs); state[1] counts builds.
// No host sync, graph-capturable.
__global__ void __launch_bounds__(256)
k_cam_key(fbbev_cam_ptrs g, int B, unsigned int* __restrict__ key, int* __restrict__ state) {
    __shared__ int differ;
    if (threadIdx.x == 0) differ = 0;
    __syncthreads();
    const int cams = B * g.N;
    const float* src[6] = {g.rots, g.trans, g.intrins, g.post_rots, g.post_trans, g.bda};
    const int len[6] = {cams * 9, cams * 3, cams * 9, cams * 9, cams * 3, B * 9};
    int off = 0, bad = 0;
    for (int s = 0; s < 6; ++s) {
        for (int i = threadIdx.x; i < len[s]; i += 256) {
            unsigned int bits;
            const float f = src[s][i];
            __builtin_memcpy(&bits, &f, 4);
            bad |= (bits != key[off + i]) ? 1 : 0;
        }
        off += len[s];
    }
    if (bad) differ = 1;
    __syncthreads();
    if (!differ) { if (threadIdx.x == 0) state[0] = 1; return; }
    off = 0;
    for (int s = 0; s < 6; ++s) {
        for (int i = threadIdx.x; i < len[s]; i += 256) {
            unsigned int bits;
            const float f = src[s][i];
            __builtin_memcpy(&bits, &f, 4);
            key[off + i] = bits;
        }
        off += len[s];
    }
    if (threadIdx.x == 0) { state[0] = 0; state[1] += 1; }
}
