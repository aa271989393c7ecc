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
               unsigned int* __restrict__ keys_out, unsigned int* __restrict__ vals_out, int* __restrict__ counts, int swz_chunks) {
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
    // swz_chunks > 0 (grid = 8 * ceil(swz_chunks / 8)): XCD-contiguous chunk order, see k_sort_scatter_seg
    int wg = (int)blockIdx.x;
    if (swz_chunks > 0) {
        wg = (int)(blockIdx.x & 7) * ((swz_chunks + 7) >> 3) + (int)(blockIdx.x >> 3);
        if (wg >= swz_chunks) return;                     // block-uniform
    }
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

// ================================================================ per-sample (segmented) sort, round 3
// The rank of a kept point of sample b lies in [b V, (b+1) V) (V = voxels per sample) whenever B V <= 2^24 -- the fp32 rank
// arithmetic of the reference is then exact -- and the points arrive in sample order.  Sorting the global keys is therefore B
// independent sorts of (key - b V) < V: 20 bits at the 200x200x16 grid instead of 24 for 16 samples, i.e. TWO passes of 10-bit
// digits instead of three of 8 (17 bits = two passes of 9 instead of three of 7 at the shipped grid).  6 launches per build
// instead of 9: keys + count matrix, scatter 0, count matrix 1, scatter 1, the two interval kernels.  Same stable order, same
// index tensors, bit for bit (tested against the oracle and against the global sort for every case the suite holds).
//   * chunks never straddle samples: workgroup w = (sample b = w / cps, chunk c = w % cps) owns points / pairs
//     [start_b + c T, min(start_b + (c+1) T, end_b)) where [start_b, end_b) is the sample's range in the pass's input order:
//     b npb .. (b+1) npb for pass 0 (all frustum points), the sample's compacted range for pass 1 -- the prefix sums of
//     ctot[] (kept points per pass-0 chunk), which every workgroup evaluates itself from the <= B cps words: no extra launch;
//   * count matrices [B cps][2^rb]; a workgroup only sums the rows of its own sample;
//   * 512-thread workgroups (8 waves x 16 rounds = 8192 pairs, two per CU): 2^10 digits x 8 waves of running counters are
//     32 KB of LDS; the per-wave offsets are formed in place.
#define FBBEV_SEG_RB 10
#define FBBEV_SEG_NB (1 << FBBEV_SEG_RB)
#define FBBEV_SEG_WAVES 8
#define FBBEV_SEG_ROUNDS 16
#define FBBEV_SEG_NT (FBBEV_SEG_WAVES * 64)
#define FBBEV_SEG_TILE (FBBEV_SEG_NT * FBBEV_SEG_ROUNDS)

struct fbbev_seg {
    int npb;              // frustum points per sample (N*D*H*W)
    int cps;              // chunks per sample = ceil(npb / FBBEV_SEG_TILE)
    unsigned int vps;     // voxels per sample (Z*Y*X): key of sample b = local key + b * vps
    int nseg;             // samples
};

// sums of ctot[0 .. lo) and ctot[0 .. hi) (lo <= hi <= nseg * cps) by the whole workgroup; red: 2 * WAVES ints of LDS
template <int WAVES>
__device__ __forceinline__ void fbbev_seg_prefix2(const int* __restrict__ ctot, int lo, int hi, int* red, int& sum_lo, int& sum_hi) {
    int a = 0, b = 0;
    for (int i = threadIdx.x; i < hi; i += WAVES * 64) {
        const int c = ctot[i];
        b += c;
        if (i < lo) a += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
    __syncthreads();
    sum_lo = 0; sum_hi = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { sum_lo += red[2 * w]; sum_hi += red[2 * w + 1]; }
    __syncthreads();
}

// keys + pass-0 count matrix + ctot, sample-aligned chunks.  GEOM: keys from the camera geometry, else from `coor`.
template <bool GEOM>
__global__ void __launch_bounds__(FBBEV_SEG_NT)
k_keys_hist_seg(fbbev_geom_src g, const float* __restrict__ coor, fbbev_grid_params gp, const float* __restrict__ depth,
                float depth_thr, fbbev_seg sg, int rb, const int* __restrict__ skip, unsigned int* __restrict__ keys_out,
                int* __restrict__ matrix, int* __restrict__ ctot) {
    if (fbbev_skip(skip)) return;
    constexpr int NT = FBBEV_SEG_NT, PER = FBBEV_SEG_ROUNDS;
    __shared__ int cnt[FBBEV_SEG_NB];
    __shared__ float m[33];
    __shared__ int tot;
    const int nb = 1 << rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    for (int d = threadIdx.x; d < nb; d += NT) cnt[d] = 0;
    if (threadIdx.x == 0) tot = 0;
    const int b = blockIdx.x / sg.cps, c = blockIdx.x - b * sg.cps;
    const int base = b * sg.npb + c * FBBEV_SEG_TILE;                       // < 2^30 (launcher)
    const int lim = (b + 1) * sg.npb;
    const int end = base + FBBEV_SEG_TILE < lim ? base + FBBEV_SEG_TILE : lim;
    const unsigned int kbase = (unsigned int)b * sg.vps;
    int mine = 0;
    if constexpr (GEOM) {
        const int dhw = g.cam.D * g.cam.H * g.cam.W;
        const int cam_lo = base / dhw, cam_hi = (end > base ? end - 1 : base) / dhw;
        constexpr int G = 8;                              // points in flight per thread (table loads of 8 points overlap; was 4)
        for (int cam = cam_lo; cam <= cam_hi; ++cam) {
            __syncthreads();
            if (threadIdx.x == 0) fbbev_cam_setup(g.cam, cam, m);
            __syncthreads();
            const int c0 = cam * dhw;
            const int lo = base > c0 ? base : c0, hi = end < c0 + dhw ? end : c0 + dhw;
            for (int r0 = 0; r0 < PER; r0 += G) {
                unsigned int key[G];
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    const int pid = base + (int)threadIdx.x + (r0 + r) * NT;
                    key[r] = (pid >= lo && pid < hi) ? fbbev_geom_key(g, m, cam, pid - c0) : FBBEV_DROP_KEY;
                }
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    const int pid = base + (int)threadIdx.x + (r0 + r) * NT;
                    if (pid >= lo && pid < hi) {
                        keys_out[pid] = key[r];
                        if (key[r] != FBBEV_DROP_KEY) { atomicAdd(&cnt[(key[r] - kbase) & dmask], 1); ++mine; }
                    }
                }
            }
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int pid = base + (int)threadIdx.x + r * NT;
            if (pid < end) {
                unsigned int key = fbbev_rank_key(coor[3 * (long long)pid], coor[3 * (long long)pid + 1], coor[3 * (long long)pid + 2], gp,
                                                  (float)b, FBBEV_DROP_KEY);
                if (depth && !(depth[pid] > depth_thr)) key = FBBEV_DROP_KEY;
                keys_out[pid] = key;
                if (key != FBBEV_DROP_KEY) { atomicAdd(&cnt[(key - kbase) & dmask], 1); ++mine; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&tot, mine);
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += NT) matrix[(long long)blockIdx.x * nb + d] = cnt[d];
    if (threadIdx.x == 0) ctot[blockIdx.x] = tot;
}

// count matrix of pass 1: chunk (b, c) of the sample's compacted range
__global__ void __launch_bounds__(FBBEV_SEG_NT)
k_sort_hist_seg(const unsigned int* __restrict__ keys, const int* __restrict__ ctot, fbbev_seg sg, int shift, int rb,
                const int* __restrict__ skip, int* __restrict__ matrix, int key_stride) {
    if (fbbev_skip(skip)) return;
    constexpr int NT = FBBEV_SEG_NT, PER = FBBEV_SEG_ROUNDS;
    __shared__ int cnt[FBBEV_SEG_NB];
    __shared__ int red[2 * FBBEV_SEG_WAVES];
    const int nb = 1 << rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    const int b = blockIdx.x / sg.cps, c = blockIdx.x - b * sg.cps;
    for (int d = threadIdx.x; d < nb; d += NT) cnt[d] = 0;
    int s0, s1;
    fbbev_seg_prefix2<FBBEV_SEG_WAVES>(ctot, b * sg.cps, (b + 1) * sg.cps, red, s0, s1);     // the sample's pairs: [s0, s1)
    const long long base = (long long)s0 + (long long)c * FBBEV_SEG_TILE;
    const unsigned int kbase = (unsigned int)b * sg.vps;
    if (base < s1) {
        unsigned int key[PER];
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const long long idx = base + threadIdx.x + r * NT;
            key[r] = (idx < s1) ? keys[idx * key_stride] : FBBEV_DROP_KEY;          // key_stride 2: (key, value) pairs
        }
#pragma unroll
        for (int r = 0; r < PER; ++r)
            if (key[r] != FBBEV_DROP_KEY) atomicAdd(&cnt[((key[r] - kbase) >> shift) & dmask], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += NT) matrix[(long long)blockIdx.x * nb + d] = cnt[d];      // zeros for an empty chunk
}

// one scatter pass of the segmented sort (pass 0 compacts and publishes P; pass 1 reads the compacted pairs)
__global__ void __launch_bounds__(FBBEV_SEG_NT)
k_sort_scatter_seg(const unsigned int* __restrict__ keys_in, const unsigned int* __restrict__ vals_in, const int* __restrict__ matrix,
                   const int* __restrict__ ctot, fbbev_seg sg, int pass, int rb, const int* __restrict__ skip,
                   unsigned int* __restrict__ keys_out, unsigned int* __restrict__ vals_out, int* __restrict__ counts, int xcd_swizzle, int pair_mode) {
    if (fbbev_skip(skip)) return;
    constexpr int WAVES = FBBEV_SEG_WAVES, ROUNDS = FBBEV_SEG_ROUNDS, NT = FBBEV_SEG_NT, NBM = FBBEV_SEG_NB;
    // chunk id of this workgroup.  xcd_swizzle (grid = 8 * ceil(chunks / 8)): workgroups are dealt to the 8 XCDs round-robin, so
    // hardware id w runs on XCD w % 8; with chunk = (w % 8) * ceil(chunks / 8) + w / 8 an XCD owns a CONTIGUOUS range of chunks
    // = whole samples.  The scatter writes 4-byte words to 2^rb digit regions; the chunks of one sample fill each region in chunk
    // order, so their partial lines meet in ONE L2 and leave it as full lines -- dealt round-robin, every XCD's L2 wrote its own
    // byte-masked fragment of every line (the passes were bound by those partial writes, not by their 30 MB of traffic).
    int chunk_id = (int)blockIdx.x;
    if (xcd_swizzle) {
        const int total = sg.nseg * sg.cps, per_xcd = (total + 7) >> 3;
        chunk_id = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
        if (chunk_id >= total) return;                    // block-uniform
    }
    __shared__ __attribute__((aligned(16))) int cnt[WAVES][NBM];      // per-wave running digit counters, then (in place) each wave's first position of a digit
    __shared__ int pos0[NBM];            // global position of this chunk's first key of a digit
    __shared__ int red[2 * WAVES];
    __shared__ int ldsw[WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = 1 << rb, shift = pass * rb;
    const unsigned int dmask = (unsigned int)nb - 1u;
    const int b = chunk_id / sg.cps, c = chunk_id - b * sg.cps;
    const unsigned int kbase = (unsigned int)b * sg.vps;
    // the sample's range in the compacted order (= where its sorted pairs go, and pass 1's input range)
    int s0, s1;
    fbbev_seg_prefix2<WAVES>(ctot, b * sg.cps, (b + 1) * sg.cps, red, s0, s1);
    if (pass == 0 && chunk_id == 0) {                    // P = all kept points
        int p0, pall;
        fbbev_seg_prefix2<WAVES>(ctot, 0, sg.nseg * sg.cps, red, p0, pall);
        if (tid == 0) counts[0] = pall;
    }
    const long long in0 = pass == 0 ? (long long)b * sg.npb : (long long)s0;                 // the sample's input range
    const long long in1 = pass == 0 ? (long long)(b + 1) * sg.npb : (long long)s1;
    const long long chunk0 = in0 + (long long)c * FBBEV_SEG_TILE;
    if (chunk0 >= in1) return;                            // block-uniform
    // the wave's pairs are requested FIRST: their round trip runs under the column sums below (round 4; the prologue used to
    // be 10.7 of the pass's 35 us with the loads behind it, profiles/r04_rank_scatter_probes.txt)
    const long long chunk = chunk0 + (long long)wave * (64 * ROUNDS);
    unsigned int k[ROUNDS], v[ROUNDS];
    int lr[ROUNDS];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long idx = chunk + r * 64 + lane;
        const bool valid = idx < in1;
        if (pair_mode & 1) {                               // uniform: the intermediate array holds (key, value) pairs
            uint2 kv = make_uint2(FBBEV_DROP_KEY, 0u);
            if (valid) kv = reinterpret_cast<const uint2*>(keys_in)[idx];
            k[r] = kv.x; v[r] = kv.y;
        } else {
            k[r] = valid ? keys_in[idx] : FBBEV_DROP_KEY;
            v[r] = valid ? (vals_in ? vals_in[idx] : (unsigned int)idx) : 0u;
        }
    }
    // column sums over the rows of this sample: digit totals and the prefix over its earlier chunks.  A thread takes FOUR
    // adjacent digits (16-byte loads) of every second row: 2 x nb / 4 threads, ceil(cps / 2) loads each in batches of 8 (two
    // round trips at cps = 31 where one digit pair per thread took eight); the two row groups meet in LDS (rows 1..4 of cnt,
    // cleared again below)
    const int row0 = b * sg.cps;
    {
        const int nq = nb >> 2;                            // nb >= 4
        const int quad = tid % nq, rg = tid / nq;          // rg < 2 works
        fbbev_v4i pre = {0, 0, 0, 0}, all = {0, 0, 0, 0};
        if (rg < 2) {
            constexpr int U = 8;
            const int* col = matrix + (long long)row0 * nb + 4 * quad;
            for (int r = rg; r < sg.cps; r += 2 * U) {
                fbbev_v4i t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int rr = r + 2 * u < sg.cps ? r + 2 * u : sg.cps - 1;          // unconditional (clamped) loads
                    __builtin_memcpy(&t[u], col + (long long)rr * nb, 16);               // 16-byte aligned: nb % 4 == 0
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + 2 * u < sg.cps) all += t[u];
                    if (r + 2 * u < c) pre += t[u];
                }
            }
            __builtin_memcpy(&cnt[1 + rg][4 * quad], &pre, 16);
            __builtin_memcpy(&cnt[3 + rg][4 * quad], &all, 16);
        }
    }
    __syncthreads();
    // exclusive scan of the digit totals over d (nb <= 2 * NT: two values per thread)
    {
        const int d0 = 2 * tid, d1 = 2 * tid + 1;
        const int a0 = d0 < nb ? cnt[3][d0] + cnt[4][d0] : 0, a1 = d1 < nb ? cnt[3][d1] + cnt[4][d1] : 0;
        const int p0 = d0 < nb ? cnt[1][d0] + cnt[2][d0] : 0, p1 = d1 < nb ? cnt[1][d1] + cnt[2][d1] : 0;
        int total;
        const int ex = fbbev_block_excl_scan_w<WAVES, false>(a0 + a1, ldsw, &total);
        if (d0 < nb) pos0[d0] = p0 + s0 + ex;
        if (d1 < nb) pos0[d1] = p1 + s0 + ex + a0;
    }
    __syncthreads();                                      // every partial has been read
    for (int i = tid; i < WAVES * NBM; i += NT) (&cnt[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool valid = k[r] != FBBEV_DROP_KEY;
        lr[r] = -1;
        const unsigned int d = ((k[r] - kbase) >> shift) & dmask;
        unsigned long long mm = __ballot(valid ? 1 : 0);
#pragma unroll
        for (int bit = 0; bit < FBBEV_SEG_RB; ++bit) {
            if (bit < rb) {
                const unsigned long long bb = __ballot((int)((d >> bit) & 1u));
                mm &= ((d >> bit) & 1u) ? bb : ~bb;
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
    for (int d = tid; d < nb; d += NT) {                 // in place: count -> first position of the wave's keys of digit d
        int run = pos0[d];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { const int t = cnt[w][d]; cnt[w][d] = run; run += t; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (lr[r] >= 0) {
            const unsigned int d = ((k[r] - kbase) >> shift) & dmask;
            const int pos = cnt[wave][d] + lr[r];
            // pair_mode & 2: ONE 8-byte store per pair into the intermediate array instead of two 4-byte stores into two arrays
            // (the scattered stores are a third of the pass: profiles/r04_rank_scatter_probes.txt)
            if (pair_mode & 2) reinterpret_cast<uint2*>(keys_out)[pos] = make_uint2(k[r], v[r]);
            else { keys_out[pos] = k[r]; vals_out[pos] = v[r]; }
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
