// history_fused_x3_kernels.h -- one history step on a 16-bit voxel-major ring as ONE kernel at fp32-grade precision (round 6):
// the trilinear warp of the T history frames, the new ring, and the two folded 1x1x1 convolutions on split operands.
//
// The two-kernel step (k_history_warp_vm, then k_history_conv_bf16x3) is a memory-bound gather kernel (3.3 ms at 400x400x16: 13 GB
// at 4 TB/s, MFMA idle) followed by an MFMA-bound one (2.9 ms, 0.57 of the MFMA issue cycles, HBM at 2.3 TB/s) that reads the
// 6.5 GB the first one just wrote.  Running them side by side on two streams does not help (profiles/r06_exp_history_step.md: a
// convolution workgroup needs a whole CU's registers and starves while warp workgroups queue).  Here one workgroup does both for a
// brick of 16 (x) by 8 (y) voxels of one z plane, every thread in two layouts:
//   * WARP layout: the brick's 1280 sixteen-byte items (voxel, 8-channel group) are dealt to the 512 threads in memory order (three
//     rounds, the last half full), so a wave's tap load and its ring store are 1 KB runs like k_history_warp_vm's.  (The first
//     form of this kernel let an MFMA lane gather the taps of its own operand -- no LDS tile, but 16 voxels x 64 bytes at a 160-byte
//     stride per instruction: four times the warp kernel's vector-L1 accesses, 4.7 G against 1.1 G, and bound by them: 6.9 ms,
//     profiles/r06_pmc_history_fused_x3_v1.json.)  Taps, weights (a per-voxel table in LDS, set up once: the flow is the
//     sample's), fma order and rounding are k_history_warp_vm's: the SAME ring bits.
//   * MFMA layout: k_history_conv_bf16x3's, one tile of 16 voxels (a brick row) per wave.  A blended item goes to the next ring
//     AND into an LDS tile [128 voxels][13 items] (double buffered) from which lane (g, j) reads channels 32 s + 8 g .. + 7 of voxel
//     j as the B operand of K step s; the T warped frames are never read back.
//   * a frame's 24 taps per thread are requested one whole FRAME before they are blended (196 KB in flight per CU against a frame's
//     75 MFMAs and ~300 VALU operations per wave); the frame's W2 / bias block comes by LDS DMA, requested at the frame's start
//     for the next one -- OLDER than the frame's tap requests, so that the barrier waits for it, not for the taps.
//     The ring stores of a frame sit between the two in issue order: every lane issues them (items outside the grid go to a
//     16-byte dump slot of the workspace), so that the count is the same in every wave and the barrier can wait with vmcnt(27).
//   * workgroups are ordered z fastest, so the 32 an XCD runs at a time are two bricks through all z planes: the y + 1 taps are
//     another wave's row (L1 / L2), the z + 1 taps another workgroup's on the same L2.
// Convolution arithmetic, fragment layouts and workspace: history_conv_x3_kernels.h (same MFMA sequence per accumulator => the
// same `out` bits as the two-kernel step).  C = Cout = 80.
#pragma once
#include "rt.h"
#include "history_kernels.h"
#include "history_conv_x3_kernels.h"

#define FBBEV_HFX_PITCH 104            // 16-bit elements of a voxel's row in the operand tile: 10 items + 2 zero items (K padding) + 1

template <int ET>
__global__ void __launch_bounds__(512, 2)
k_history_fused_x3(const void* __restrict__ hist, long long hist_stride_b, void* __restrict__ nxt, long long nxt_stride_b,
                   const float* __restrict__ flow, const unsigned short* __restrict__ w1x, const float* __restrict__ biasx,
                   const unsigned short* __restrict__ w2x, const float* __restrict__ bias2, int T1, int Z, int Y, int X,
                   int n_xt, int n_yt, int per_xcd, int n_work, float* __restrict__ out, void* __restrict__ dump) {
    static_assert(ET == 1 || ET == 2, "16-bit voxel-major ring");
    constexpr int MT1 = 5, MT2 = 5, NT = 512, VOX = 128, NR = 3;
    constexpr int C = 16 * MT1, Cout = 16 * MT2, KS = (C + 31) / 32, PV = C / 8, NI = VOX * PV;
    constexpr int A1 = MT1 * KS * 64 * 8, A2 = MT2 * KS * 64 * 8;
    constexpr int NW2 = 2 * A2 / 8, NBI = C / 4, NP = NW2 + NBI, A2P = (NP + NT - 1) / NT, A2S = A2P * NT * 8;
    constexpr int NW1 = 2 * A1 / 8, IW1 = (NW1 + NT - 1) / NT;
    constexpr int XP = FBBEV_HFX_PITCH, XT = VOX * XP;
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());
    unsigned short* a2buf = lds;                          // [2][A2S]: W2_t (hi | lo) and the frame's bias
    unsigned short* w1buf = lds + 2 * A2S;                // [hi A1 | lo A1]
    unsigned short* xtile = w1buf + 2 * A1;               // [2][VOX][XP]: the frame's operands, a row per voxel
    unsigned int* geo = reinterpret_cast<unsigned int*>(xtile + 2 * XT);   // [VOX][16]: 8 tap row offsets (bytes), 8 tap weights
    int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);            // one contiguous eighth per XCD
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= n_work) return;
    const int z = work % Z; work /= Z;
    const int xt = work % n_xt; work /= n_xt;
    const int yt = work % n_yt, b = work / n_yt;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, j = lane & 15;
    const long long N = (long long)Z * Y * X;
    const size_t frame_bytes = (size_t)N * C * 2;
    const char* src = static_cast<const char*>(hist) + (size_t)b * hist_stride_b * 2;
    char* dstb = static_cast<char*>(nxt) + (size_t)b * nxt_stride_b * 2;
    // ---- WARP layout: item i = tid + 512 r = (voxel i / 10 of the brick, channel group i % 10)
    unsigned int own[NR];        // the item's byte offset in a frame (0 for an item outside the grid / beyond the brick's 1280)
    unsigned int gofs[NR];       // bits 0-11: the voxel's entry in the table (dwords); 12-15: the channel group; 16-31: place in the operand tile (elements)
    bool live[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = tid + NT * r;
        const int vl = i < NI ? i / PV : 0, p = i < NI ? i % PV : 0;
        const int x = xt * 16 + (vl & 15), y = yt * 8 + (vl >> 4);
        live[r] = i < NI && x < X && y < Y;
        own[r] = live[r] ? (unsigned int)((((long long)(z * Y + y) * X + x) * C + 8 * p) * 2) : 0u;
        gofs[r] = (unsigned int)(vl * 16) | ((unsigned int)p << 12) | ((unsigned int)(vl * XP + 8 * p) << 16);
    }
    // source of piece i = thread + NT q of a frame's block: W2_t (hi | lo), then the frame's (scaled) bias; past the end: any valid piece
    auto block_src = [&](int t, int q) -> const fbbev_v4u* {
        const int i = tid + NT * q;
        const fbbev_v4u* wsrc = reinterpret_cast<const fbbev_v4u*>(w2x + (long long)t * 2 * A2);
        const fbbev_v4u* bsrc = reinterpret_cast<const fbbev_v4u*>(biasx + ((long long)b * T1 + t) * C);
        if (NT * (q + 1) <= NW2) return wsrc + i;
        return i < NW2 ? wsrc + i : bsrc + (i - NW2 < NBI ? i - NW2 : 0);
    };
    {   // block 0, W1 (hi | lo) and the current frame's items (frame 0: taken as stored): requested together
        fbbev_v4u ta[A2P], tb[IW1], tc[NR];
#pragma unroll
        for (int q = 0; q < A2P; ++q) ta[q] = *block_src(0, q);
#pragma unroll
        for (int k = 0; k < IW1; ++k) { const int i = tid + NT * k; tb[k] = reinterpret_cast<const fbbev_v4u*>(w1x)[i < NW1 ? i : 0]; }
#pragma unroll
        for (int r = 0; r < NR; ++r) tc[r] = *reinterpret_cast<const fbbev_v4u*>(dstb + own[r]);
        // the voxel table: thread v < 128 sets up voxel v (k_history_warp's expression sequence: fbbev_warp_taps)
        if (tid < VOX) {
            const int x = xt * 16 + (tid & 15), y = yt * 8 + (tid >> 4);
            int tv[8];
            float w[8];
            if (x < X && y < Y) fbbev_warp_taps(flow + b * 16, x, y, z, X, Y, Z, tv, w);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { tv[k] = 0; w[k] = 0.f; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                geo[tid * 16 + k] = (unsigned int)tv[k] * (unsigned int)(C * 2);
                unsigned int wb;
                __builtin_memcpy(&wb, &w[k], 4);
                geo[tid * 16 + 8 + k] = wb;
            }
        }
        // the operand tiles' K padding (items 10, 11 of every row of both buffers) is zero and stays zero
        for (int i = tid; i < 2 * VOX; i += NT) {
            fbbev_v4u* row = reinterpret_cast<fbbev_v4u*>(xtile + (size_t)i * XP);
            row[PV] = fbbev_v4u{0u, 0u, 0u, 0u};
            row[PV + 1] = fbbev_v4u{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int q = 0; q < A2P; ++q) reinterpret_cast<fbbev_v4u*>(a2buf)[tid + NT * q] = ta[q];
#pragma unroll
        for (int k = 0; k < IW1; ++k) { const int i = tid + NT * k; if (i < NW1) reinterpret_cast<fbbev_v4u*>(w1buf)[i] = tb[k]; }
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (tid + NT * r < NI) *reinterpret_cast<fbbev_v4u*>(xtile + (gofs[r] >> 16)) = tc[r];
    }
    __syncthreads();                                        // the voxel table is read by other threads than wrote it
    fbbev_v4f acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
    // tap[r]: the 8 taps of item r of the NEXT frame to blend, RAW.  The requests are the youngest 24 loads of the wave whenever it
    // reaches the frame barrier (the DMA of the frame's block and the ring stores are older).
    fbbev_v4u tap[NR][8];
    auto request = [&](int hf) {                            // the items of the warp of history frame hf
        const char* nsrc = src + (size_t)hf * frame_bytes;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const unsigned int* ge = geo + (gofs[r] & 0xfffu);
            const fbbev_v4u o0 = *reinterpret_cast<const fbbev_v4u*>(ge), o1 = *reinterpret_cast<const fbbev_v4u*>(ge + 4);
            const unsigned int po = ((gofs[r] >> 12) & 0xfu) * 16u;
#pragma unroll
            for (int k = 0; k < 8; ++k) tap[r][k] = *reinterpret_cast<const fbbev_v4u*>(nsrc + ((k < 4 ? o0[k] : o1[k - 4]) + po));
        }
    };
    request(0);
    const fbbev_v4f zero4f = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T1; ++t) {
        // block t has landed when everything older than the wave's 24 tap requests and the NR ring stores in front of them has:
        // waiting for the stores too (vmcnt(24)) put a store round trip into every frame
        fbbev_wait_loads_but<8 * NR + NR>();
        __syncthreads();                                        // block t in a2buf[t & 1], frame t's operands in xtile[t & 1]
        const unsigned short* a2t = a2buf + (t & 1) * A2S;
        const unsigned short* xr = xtile + (t & 1) * XT + (wave * 16 + j) * XP + 8 * g;
        {   // the next frame's block, straight into the LDS buffer the barrier just freed (older than this frame's tap requests)
            const int tn = t + 1 < T1 ? t + 1 : t;
            unsigned short* nb = a2buf + ((t + 1) & 1) * A2S;
#pragma unroll
            for (int q = 0; q < A2P; ++q) fbbev_lds_dma16(block_src(tn, q), nb + (size_t)(wave * 64 + NT * q) * 8);
        }
        // ---- convolution 1: acc1 starts at the frame's bias; B operand of K step s = item 4 s + g of voxel j
        fbbev_v4f acc1[MT1];
        {
            const float* bt = reinterpret_cast<const float*>(a2t + 2 * A2);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc1[mt] = *reinterpret_cast<const fbbev_v4f*>(bt + 16 * mt + 4 * g);
        }
        fbbev_v4u xp[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xp[s] = *reinterpret_cast<const fbbev_v4u*>(xr + 32 * s);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            fbbev_v4u af[MT1];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) af[mt] = *reinterpret_cast<const fbbev_v4u*>(w1buf + A1 + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc1[mt] = fbbev_mfma_16x16x32_raw<ET>(af[mt], xp[s], acc1[mt]);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) af[mt] = *reinterpret_cast<const fbbev_v4u*>(w1buf + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc1[mt] = fbbev_mfma_16x16x32_raw<ET>(af[mt], xp[s], acc1[mt]);
        }
        // issue order of the 30 fragment reads and 30 MFMAs above: five reads ahead, then one read behind every MFMA (into the
        // registers that MFMA just consumed); the region's first MT1 + KS LDS reads are the bias and the operands
        FBBEV_SCHED_LDS_READ(2 * MT1 + KS);
#pragma unroll
        for (int i = 0; i < 2 * KS * MT1 - MT1; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }
        FBBEV_SCHED_MFMA(MT1);
        fbbev_sched_fence();
        // ---- convolution 2: acc2 += W2'_t . y', y' = relu(acc1) split into bf16 hi / lo in the K order of W2's fragments (step s =
        // accumulator tiles 2 s, 2 s + 1), one K step at a time
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            fbbev_bf16x8 yh, yl;
            {
                fbbev_v4f y0, y1 = zero4f;
#pragma unroll
                for (int r = 0; r < 4; ++r) y0[r] = fmaxf(acc1[2 * s][r], 0.f);
                if (2 * s + 1 < MT1) {
                    const int m1 = 2 * s + 1 < MT1 ? 2 * s + 1 : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y1[r] = fmaxf(acc1[m1][r], 0.f);
                }
                fbbev_split_bf16x8(y0, y1, yh, yl);
            }
            fbbev_bf16x8 af[MT2];
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) af[mt] = fbbev_ld_bf16x8(a2t + A2 + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yh, acc2[mt]);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) af[mt] = fbbev_ld_bf16x8(a2t + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yl, acc2[mt]);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yh, acc2[mt]);
        }
        // issue order of the 30 fragment reads and 45 MFMAs above (per K step: lo fragments x 5 MFMAs, hi fragments x 10): a read goes
        // into the registers of the MFMA just issued when that was the fragment's last use
        FBBEV_SCHED_LDS_READ(MT2);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < MT2; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }        // lo . yh; the hi fragments behind them
            FBBEV_SCHED_MFMA(MT2);                                                               // hi . yl
            if (s + 1 < KS) {
#pragma unroll
                for (int i = 0; i < MT2; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }    // hi . yh; the next step's lo fragments
            } else FBBEV_SCHED_MFMA(MT2);
        }
        fbbev_sched_fence();
        // ---- WARP layout: frame t + 1 = the warp of history frame t, blended from the taps requested a frame ago; to the next ring
        // (slot t + 1) and into the other operand tile
        if (t + 1 < T1) {
            char* fdst = dstb + (size_t)(t + 1) * frame_bytes;
            unsigned short* xw = xtile + ((t + 1) & 1) * XT;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const unsigned int* ge = geo + (gofs[r] & 0xfffu) + 8;
                const fbbev_v4f w0 = *reinterpret_cast<const fbbev_v4f*>(ge), w1 = *reinterpret_cast<const fbbev_v4f*>(ge + 4);
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float wk = k < 4 ? w0[k] : w1[k - 4];
                    if constexpr (ET == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[2 * e] = fbbev_fma_f16<0>(tap[r][k][e], wk, acc[2 * e]);
                            acc[2 * e + 1] = fbbev_fma_f16<1>(tap[r][k][e], wk, acc[2 * e + 1]);
                        }
                    } else {
                        float a[8];
                        fbbev_widen_vec<ET>(tap[r][k], a);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], wk, acc[e]);
                    }
                }
                const fbbev_v4u item = fbbev_narrow_vec<ET>(acc);
                // EVERY lane stores (an item outside the grid goes to a dump slot): the count of stores between the block's DMA and
                // the tap requests must not depend on the lane mask (a wave without live items would skip the instruction)
                *reinterpret_cast<fbbev_v4u*>(live[r] ? fdst + own[r] : static_cast<char*>(dump)) = item;
                if (tid + NT * r < NI) *reinterpret_cast<fbbev_v4u*>(xw + (gofs[r] >> 16)) = item;
            }
        }
        fbbev_sched_fence();
        // the items of frame t + 2 (the warp of history frame t + 1; clamped at the end: the last requests are not used)
        request(t + 1 < T1 - 1 ? t + 1 : T1 - 2);
    }
    const int x = xt * 16 + j, y = yt * 8 + wave;
    if (x < X && y < Y) {
        float* ob_ = out + (long long)b * Cout * N + ((long long)(z * Y + y) * X + x);
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob_[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
    }
}
