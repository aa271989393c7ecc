// wgrad_kernels.h -- weight and bias gradients of the row-wise linear layers of the backward projection (training, round 6):
//     gW (O, I) = gy^T x = sum over the R rows of gy[r, :]^T x[r, :],        gb (O) = sum over the rows of gy[r, :]
// for gy (R, O), x (R, I) with R = 160 000 BEV-query rows (BASELINE configs[2], B = 4) and O x I between 32 x 80 and 512 x 80.
// Reference: autograd's mm(gy^T, x) / gy.sum(0) behind every nn.Linear of bevformer_encoder.py:206-377 and
// spatial_cross_attention_depth.py:432-436,464.  Until round 6 this was a split-K batched vendor GEMM + a ones-vector GEMM + three
// ATen reductions per layer: 1.2 ms of Cijk_* kernels + 0.3 ms of reduce kernels per step for ~2 GB of operands.
//
// Arithmetic: the split-operand bf16 MFMA of rows_linear_kernels.h (v = hi + lo, three MFMAs per product, fp32 accumulation,
// ~1e-5 relative), K = the rows.  Both operands are row-major with the reduction index r as the SLOW index, while an MFMA lane wants
// 8 consecutive k of one column: every 32-row step is transposed through LDS -- a thread loads two consecutive rows x four columns
// (two coalesced 16-byte loads), splits them, and writes per column ONE dword holding the (row, row + 1) pair of hi (lo) halves
// straight into the fragment order [tile of 16 columns][hi | lo][lane = (k-block, column)][8 bf16]; a tile is padded by 2 dwords
// so that the 32 column quads of a wave's writes fall into 32 different bank pairs (fragments are then read as two 8-byte pieces).
// Double-buffered: the global loads of step s + 1 are requested before the MFMAs of step s, one barrier per step.
// Split-K over workgroups with per-workgroup partial results and a fixed-order reduction (k_rows_wgrad_reduce): no atomics,
// bit-identical run to run.  Bound: HBM (each operand read once per 128-wide output chunk).
#pragma once
#include "rt.h"
#include "x3_split.h"

#define FBBEV_WG_TILE_DW 514      // dwords of a 16-column tile in LDS: hi block (64 lanes x 4 dwords), lo block, 2 dwords of padding

// (a, b) = the values of rows (2p, 2p + 1) of one column -> packed hi pair, packed lo pair (element 2p in the low half)
__device__ __forceinline__ void fbbev_wg_split_pair(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = fbbev_cvt_pk16<1>(a, b);
    const unsigned int ua = hi << 16, ub = hi & 0xffff0000u;
    float fa, fb;
    __builtin_memcpy(&fa, &ua, 4); __builtin_memcpy(&fb, &ub, 4);
    lo = fbbev_cvt_pk16<1>(a - fa, b - fb);
}

__device__ __forceinline__ fbbev_bf16x8 fbbev_wg_frag(const unsigned int* p) {          // 16 bytes at an 8-byte aligned LDS address
    const fbbev_v4f v = fbbev_lds_ld_v4f_a8(reinterpret_cast<const float*>(p));
    fbbev_bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// grid = n_split * n_oc * n_ic workgroups of 256 threads; workgroup (split, oc, ic) accumulates output rows [128 oc, 128 oc + 128) x
// input columns [16 NTI ic, 16 NTI (ic + 1)) over the 32-row steps [split * kps, (split + 1) * kps) and writes
// part_w[split][o][i] (dense (O, I) per split); the ic == 0 workgroups also write part_b[split][o] when part_b is given.
// addend (period, I), optional: the layer's input rows were x[r] + addend[r % period] (query + query_pos, never materialised).
template <int NTI, bool ADD>
__global__ void __launch_bounds__(256, 2)
k_rows_wgrad_x3(const float* __restrict__ gy, long long ldg, const float* __restrict__ x, long long ldx,
                const float* __restrict__ addend, long long ld_add, int add_period, long long rows, int O, int I,
                int n_oc, int n_ic, int ksteps, int kps, int amt, float* __restrict__ part_w, float* __restrict__ part_b) {
    constexpr int XQ = 4 * NTI;
    // amt = 16-column tiles of grad_out a workgroup stages (min(8, O / 16 rounded up)): a narrow layer takes less LDS -> three workgroups per CU
    const int BUF = (amt + NTI) * FBBEV_WG_TILE_DW;
    unsigned int* lds = reinterpret_cast<unsigned int*>(fbbev_dyn_lds_f32());               // [2][amt + NTI tiles][TILE_DW]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, g = lane >> 4, j = lane & 15;
    int bid = blockIdx.x;
    const int ic = bid % n_ic; bid /= n_ic;
    const int oc = bid % n_oc;
    const int split = bid / n_oc;
    const int o0 = oc * 128, i0 = ic * 16 * NTI;
    const int nmt = (O - o0 >= 128) ? 8 : (O - o0 + 15) / 16;
    const int ks0 = split * kps, ks1 = ks0 + kps < ksteps ? ks0 + kps : ksteps;
    // this thread's two items of each operand: item = t + 256 u -> (row pair, column quad)
    const int gcq = t & 31, grp0 = t >> 5;                                                  // gy: 16 row pairs x 32 column quads
    const bool gcol = o0 + 4 * gcq < O;
    int xcq[2], xrp[2];
    bool xcol[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = t + 256 * u;
        xcq[u] = it % XQ; xrp[u] = it / XQ;                                                 // x: 16 row pairs x XQ column quads
        xcol[u] = xrp[u] < 16 && i0 + 4 * xcq[u] < I;
    }
    // a step's pieces stay RAW in registers from `request` to `commit` (round-6 lesson, as in rows_linear_kernels.h: a select or
    // an add on a loaded value right behind its load put an s_waitcnt vmcnt(0) after every piece -- four round trips per step in
    // front of the MFMAs instead of one behind them); out-of-range pieces load a clamped address and are zeroed at commit
    fbbev_v4f ga[2][2], xa[2][2], va[ADD ? 2 : 1][2];
    const fbbev_v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    // row of the addend for this thread's x rows: (row % period), carried from step to step (steps are requested in ascending order)
    int ra[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) ra[u][h] = ADD ? (int)(((long long)ks0 * 32 + 2 * xrp[u] + h) % add_period) : 0;
    const int step_mod = ADD ? 32 % add_period : 0;
    auto request = [&](int ks) {
        const long long r0 = (long long)ks * 32;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long rg = r0 + 2 * (grp0 + 8 * u), rx = r0 + 2 * xrp[u];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool okg = gcol && rg + h < rows, okx = xcol[u] && rx + h < rows;
                ga[u][h] = *reinterpret_cast<const fbbev_v4f*>(gy + (okg ? (rg + h) * ldg + o0 + 4 * gcq : 0));
                xa[u][h] = *reinterpret_cast<const fbbev_v4f*>(x + (okx ? (rx + h) * ldx + i0 + 4 * xcq[u] : 0));
                if constexpr (ADD) {
                    va[u][h] = *reinterpret_cast<const fbbev_v4f*>(addend + (okx ? (long long)ra[u][h] * ld_add + i0 + 4 * xcq[u] : 0));
                    ra[u][h] += step_mod;
                    if (ra[u][h] >= add_period) ra[u][h] -= add_period;
                }
            }
        }
    };
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto commit = [&](unsigned int* buf, int ks) {
        const long long r0 = (long long)ks * 32;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if ((gcq >> 2) < amt) {
                const int rp = grp0 + 8 * u;
                const long long rg = r0 + 2 * rp;
                const fbbev_v4f a = (gcol && rg < rows) ? ga[u][0] : zero4, b = (gcol && rg + 1 < rows) ? ga[u][1] : zero4;
                unsigned int* dst = buf + (gcq >> 2) * FBBEV_WG_TILE_DW + ((rp >> 2) * 16 + 4 * (gcq & 3)) * 4 + (rp & 3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned int hi, lo;
                    fbbev_wg_split_pair(a[k], b[k], hi, lo);
                    dst[4 * k] = hi; dst[4 * k + 256] = lo;
                    bsum[k] += a[k] + b[k];
                }
            }
            if (xrp[u] < 16) {
                const int rp = xrp[u];
                const long long rx = r0 + 2 * rp;
                fbbev_v4f a = xa[u][0], b = xa[u][1];
                if constexpr (ADD) { a = a + va[u][0]; b = b + va[u][1]; }
                a = (xcol[u] && rx < rows) ? a : zero4; b = (xcol[u] && rx + 1 < rows) ? b : zero4;
                unsigned int* dst = buf + (amt + (xcq[u] >> 2)) * FBBEV_WG_TILE_DW + ((rp >> 2) * 16 + 4 * (xcq[u] & 3)) * 4 + (rp & 3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned int hi, lo;
                    fbbev_wg_split_pair(a[k], b[k], hi, lo);
                    dst[4 * k] = hi; dst[4 * k + 256] = lo;
                }
            }
        }
    };
    fbbev_v4f acc[2][NTI];
#pragma unroll
    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) acc[ml][nt] = zero4;
    if (ks0 < ks1) {
        request(ks0);
        commit(lds, ks0);
    }
    __syncthreads();
    for (int ks = ks0; ks < ks1; ++ks) {
        const unsigned int* cur = lds + ((ks - ks0) & 1) * BUF;
        unsigned int* nxt = lds + (((ks - ks0) & 1) ^ 1) * BUF;
        const bool more = ks + 1 < ks1;                                                     // uniform
        if (more) request(ks + 1);
        fbbev_sched_fence();                                                                // the requests stay ahead of the MFMAs
        fbbev_bf16x8 ah[2], al[2];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            const unsigned int* base = cur + (2 * wave + ml) * FBBEV_WG_TILE_DW + lane * 4;
            ah[ml] = fbbev_wg_frag(base); al[ml] = fbbev_wg_frag(base + 256);
        }
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) {
            if (i0 + 16 * nt >= I) break;                                                   // uniform
            const unsigned int* base = cur + (amt + nt) * FBBEV_WG_TILE_DW + lane * 4;
            const fbbev_bf16x8 bh = fbbev_wg_frag(base), bl = fbbev_wg_frag(base + 256);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) {
                if (2 * wave + ml >= nmt) break;                                            // uniform
                acc[ml][nt] = fbbev_mfma_f32_16x16x32_bf16(al[ml], bh, acc[ml][nt]);
                acc[ml][nt] = fbbev_mfma_f32_16x16x32_bf16(ah[ml], bl, acc[ml][nt]);
                acc[ml][nt] = fbbev_mfma_f32_16x16x32_bf16(ah[ml], bh, acc[ml][nt]);
            }
        }
        fbbev_sched_fence();
        if (more) commit(nxt, ks + 1);
        __syncthreads();
    }
    // accumulator register r of tile (ml, nt) = gW[o0 + 16 (2 wave + ml) + 4 g + r][i0 + 16 nt + j]
    float* pw = part_w + (long long)split * O * I;
#pragma unroll
    for (int ml = 0; ml < 2; ++ml) {
        if (2 * wave + ml >= nmt) break;
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) {
            const int i = i0 + 16 * nt + j;
            if (i0 + 16 * nt >= I) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * (2 * wave + ml) + 4 * g + r;
                if (o < O && i < I) pw[(long long)o * I + i] = acc[ml][nt][r];
            }
        }
    }
    if (part_b && ic == 0) {                                                                // uniform
        float* red = reinterpret_cast<float*>(lds);                                         // [8 row-pair groups][128 columns]
#pragma unroll
        for (int k = 0; k < 4; ++k) red[grp0 * 128 + 4 * gcq + k] = bsum[k];
        __syncthreads();
        if (t < 128 && o0 + t < O) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * 128 + t];
            part_b[(long long)split * O + o0 + t] = s;
        }
    }
}

// gw[idx] = sum over the splits of part_w[split][idx] (idx < OI), gb[o] likewise from part_b: a workgroup = 32 outputs x LANES split lanes,
// each lane adds its splits in ascending order, the lane sums are added in lane order -- one fixed association for every launch
template <int LANES>
__global__ void __launch_bounds__(32 * LANES)
k_rows_wgrad_reduce(const float* __restrict__ part_w, const float* __restrict__ part_b, int n_split, long long OI, int O,
                    float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[LANES][32];
    const int ii = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long long idx = (long long)blockIdx.x * 32 + ii;
    const long long total = OI + (gb ? O : 0);
    float s = 0.f;
    if (idx < total) {
        const float* src = idx < OI ? part_w + idx : part_b + (idx - OI);
        const long long stride = idx < OI ? OI : (long long)O;
        for (int sp = sl; sp < n_split; sp += LANES) s += src[(long long)sp * stride];
    }
    red[sl][ii] = s;
    __syncthreads();
    if (sl == 0 && idx < total) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < LANES; ++q) r += red[q][ii];
        if (idx < OI) gw[idx] = r; else gb[idx - OI] = r;
    }
}

// out[n] = sum over b < B of x[b][n] (ascending b): the batch sum behind a parameter that every sample shares (the positional table
// under `query + query_pos`, the BEV embedding) -- ATen's reduction over the leading dimension runs at ~0.6 TB/s on (4, 3.2 M)
template <int UNUSED>
__global__ void __launch_bounds__(256)
k_sum_leading(const float* __restrict__ x, const float* __restrict__ x2, int B, long long N4, fbbev_v4f* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N4) return;
    fbbev_v4f s = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        s = s + reinterpret_cast<const fbbev_v4f*>(x)[(long long)b * N4 + i];
        if (x2) s = s + reinterpret_cast<const fbbev_v4f*>(x2)[(long long)b * N4 + i];
    }
    out[i] = s;
}

// softmax over groups of N consecutive floats (N in {4, 8, 16, 32}: the L * P attention logits of one (query, head)) and its backward
//   y = softmax(x),      gx = y * (gy - sum(y * gy))        (gx may alias gy)
// a lane holds four consecutive floats, N / 4 lanes a group (xor-shuffles inside the group): ATen's softmax_backward_data first
// materialises gy * y in a pass of its own (0.08 ms of the 0.17 ms at 160 000 x 8 groups of 32) and both directions run at ~3.5 TB/s.
template <int N, bool BWD>
__global__ void __launch_bounds__(256)
k_softmax_groups(const float* __restrict__ a, const float* b, long long n4, float* out) {
    constexpr int LPG = N / 4;                                       // lanes per group
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n4;
    const fbbev_v4f va = *reinterpret_cast<const fbbev_v4f*>(a + 4 * (live ? i : 0));
    if constexpr (!BWD) {
        float mx = fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3]));
#pragma unroll
        for (int o = LPG >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        fbbev_v4f e;
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = __expf(va[k] - mx);
        float sm = (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
        for (int o = LPG >> 1; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float inv = 1.0f / sm;
        if (live) *reinterpret_cast<fbbev_v4f*>(out + 4 * i) = e * inv;
    } else {
        const fbbev_v4f vg = *reinterpret_cast<const fbbev_v4f*>(b + 4 * (live ? i : 0));      // a = y, b = gy
        float dot = (va[0] * vg[0] + va[1] * vg[1]) + (va[2] * vg[2] + va[3] * vg[3]);
#pragma unroll
        for (int o = LPG >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        fbbev_v4f r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = va[k] * (vg[k] - dot);
        if (live) *reinterpret_cast<fbbev_v4f*>(out + 4 * i) = r;
    }
}
