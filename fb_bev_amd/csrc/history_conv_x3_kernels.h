// history_conv_x3_kernels.h -- the two folded 1x1x1 convolutions of the temporal fusion at fp32-GRADE precision on the bf16
// MFMA ("bf16 x 3", round 3).
//
// The exact route (k_history_conv_t: v_mfma_f32_16x16x4_f32) is compute bound: 10.5 ms of the 16.6 ms BASELINE configs[4] step.
// The bf16 route (k_history_conv_bf16) is 5x faster but rounds weights, frames and the intermediate to 8 mantissa bits (~4e-3
// of the output peak).  Here every operand is split into TWO bf16 terms, v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (16 mantissa bits), and a product is three MFMAs:  a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo  (the dropped lo.lo term and the
// split's own remainder are both ~2^-17 relative), fp32 accumulation as before:
//   * an fp16 ring element has 11 mantissa bits: its split is EXACT; a bf16 ring element is its own hi (two MFMAs);
//   * the folded weights are split once per launch (k_history_weight_fragments_bf16x3), the ReLU'd intermediate when it is
//     parked in LDS (two rows per voxel: hi and lo).
// Error against the fp32 convolutions of the same stored frames: ~1e-5 of the output peak (tests) -- three decimal digits
// better than the TF32 arithmetic PyTorch's cuDNN convolutions use by default on the reference's own hardware, at ~3x the
// bf16 kernel's MFMA work instead of the fp32 MFMA's 16x.
// Shape: voxel-major 16-bit ring only.  512 threads = 8 waves x 16 voxels per workgroup so that the LDS-staged weights (W2 hi + lo,
// 30 KB per frame, double buffered; W1 lo 15 KB; W1 hi stays in registers) are shared by 128 voxels: 132 KB, one workgroup per CU.
#pragma once
#include "rt.h"
#include "history_kernels.h"
#include "history_conv_kernels.h"

#include "x3_split.h"

template <int MT1, int MT2, int ET>
__global__ void __launch_bounds__(512)
k_history_conv_bf16x3(const void* __restrict__ feats, long long fstride_b, const unsigned short* __restrict__ w1x,
                      const float* __restrict__ bias1, const unsigned short* __restrict__ w2x, const float* __restrict__ bias2,
                      int T1, int N, int tiles_per_b, float* __restrict__ out) {
    static_assert(ET == 1 || ET == 2, "16-bit voxel-major ring");
    constexpr int C = 16 * MT1, Cout = 16 * MT2, KS = (C + 31) / 32, CP = KS * 32, PITCH = CP + 8;
    constexpr int A1 = MT1 * KS * 64 * 8;                 // bf16 elements of one W1 part (hi or lo)
    constexpr int A2 = MT2 * KS * 64 * 8;                 // ... of one W2 part of one frame; a frame's block is [hi | lo]
    constexpr int A2P = (2 * A2 / 8 + 511) / 512;         // 16-byte pieces of a frame's block per thread
    constexpr int A2S = A2P * 512 * 8;                    // elements of a staging buffer
    constexpr int PF = 3;                                 // frames of X in flight
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());
    unsigned short* a2buf = lds;                          // [2][A2S]
    unsigned short* a1lo = lds + 2 * A2S;                 // [A1]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int n = tile * 128 + wave * 16 + j;
    const bool inb = n < N;
    unsigned short* yh = a1lo + A1 + ((wave * 16 + j) * 2) * PITCH;      // this voxel's hi row; the lo row follows it
    unsigned short* yl = yh + PITCH;
    for (int c = C + g; c < CP; c += 4) { yh[c] = 0; yl[c] = 0; }        // padding channels of the intermediate
    for (int i = threadIdx.x; i < 2 * A2 / 8; i += 512)                   // W2_0 (hi | lo)
        reinterpret_cast<fbbev_v4u*>(a2buf)[i] = reinterpret_cast<const fbbev_v4u*>(w2x)[i];
    for (int i = threadIdx.x; i < A1 / 8; i += 512)                       // W1 lo
        reinterpret_cast<fbbev_v4u*>(a1lo)[i] = reinterpret_cast<const fbbev_v4u*>(w1x + A1)[i];
    const long long xb = (long long)b * fstride_b;
    fbbev_bf16x8 a1[MT1][KS];
#pragma unroll
    for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int s = 0; s < KS; ++s) a1[mt][s] = fbbev_ld_bf16x8(w1x + ((mt * KS + s) * 64 + lane) * 8);
    fbbev_v4f acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
    fbbev_v4u bv[PF][KS];                                   // X stays RAW in registers until it is used
    unsigned int xoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        xoff[s] = (inb && c < C) ? (unsigned int)(((long long)n * C + c) * 2) : 0u;
    }
    auto load_x = [&](int slot, long long base) {
        const char* fb = static_cast<const char*>(feats) + base * 2;
#pragma unroll
        for (int s = 0; s < KS; ++s) bv[slot][s] = *reinterpret_cast<const fbbev_v4u*>(fb + xoff[s]);
    };
    const long long fsz = (long long)C * N;
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (u < T1) load_x(u, xb + (long long)u * fsz);
    const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
    auto frame = [&](int t, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        __syncthreads();                                        // W2_t is in a2buf[t & 1]; a2buf[(t + 1) & 1] is free again
        const float* b1 = bias1 + ((long long)b * T1 + t) * C;
        fbbev_v4u wst[A2P];
        {
            const fbbev_v4u* wn = reinterpret_cast<const fbbev_v4u*>(w2x + (long long)(t + 1 < T1 ? t + 1 : t) * 2 * A2);
#pragma unroll
            for (int q = 0; q < A2P; ++q) {
                const int i = threadIdx.x + 512 * q;
                wst[q] = wn[i < 2 * A2 / 8 ? i : 0];
            }
        }
        fbbev_sched_fence();
        fbbev_v4f bia[MT1], acc1[MT1];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            bia[mt] = *reinterpret_cast<const fbbev_v4f*>(b1 + 16 * mt + 4 * g);
            acc1[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        }
        fbbev_sched_fence();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = inb && 32 * s + 8 * g < C;
            fbbev_bf16x8 xh, xl = zero8;
            if constexpr (ET == 1) {
                __builtin_memcpy(&xh, &bv[SL][s], 16);           // a bf16 ring piece is its own hi term
            } else {
                fbbev_v4f lo4, hi4;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    lo4[2 * e] = fbbev_widen<ET>(bv[SL][s][e] & 0xffffu);     lo4[2 * e + 1] = fbbev_widen<ET>(bv[SL][s][e] >> 16);
                    hi4[2 * e] = fbbev_widen<ET>(bv[SL][s][2 + e] & 0xffffu); hi4[2 * e + 1] = fbbev_widen<ET>(bv[SL][s][2 + e] >> 16);
                }
                fbbev_split_bf16x8(lo4, hi4, xh, xl);            // exact: 11 mantissa bits = 8 + 3
                xl = ok ? xl : zero8;
            }
            xh = ok ? xh : zero8;
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(a1lo + ((mt * KS + s) * 64 + lane) * 8);
                acc1[mt] = fbbev_mfma_f32_16x16x32_bf16(al, xh, acc1[mt]);
                if constexpr (ET != 1) acc1[mt] = fbbev_mfma_f32_16x16x32_bf16(a1[mt][s], xl, acc1[mt]);
                acc1[mt] = fbbev_mfma_f32_16x16x32_bf16(a1[mt][s], xh, acc1[mt]);
            }
        }
        fbbev_sched_fence();
        load_x(SL, xb + (long long)(t + PF < T1 ? t + PF : T1 - 1) * fsz);
        fbbev_sched_fence();
        fbbev_wave_sync();                                                  // the Y rows are wave-private
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            const fbbev_v4f y = {fmaxf(acc1[mt][0] + bia[mt][0], 0.f), fmaxf(acc1[mt][1] + bia[mt][1], 0.f),
                                 fmaxf(acc1[mt][2] + bia[mt][2], 0.f), fmaxf(acc1[mt][3] + bia[mt][3], 0.f)};
            fbbev_bf16x8 h8, l8;
            fbbev_split_bf16x8(y, y, h8, l8);
            unsigned long long fh, fl;
            __builtin_memcpy(&fh, &h8, 8);
            __builtin_memcpy(&fl, &l8, 8);
            *reinterpret_cast<unsigned long long*>(yh + 16 * mt + 4 * g) = fh;
            *reinterpret_cast<unsigned long long*>(yl + 16 * mt + 4 * g) = fl;
        }
        fbbev_wave_sync();
        const unsigned short* a2t = a2buf + (t & 1) * A2S;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const fbbev_bf16x8 yho = fbbev_ld_bf16x8(yh + 32 * s + 8 * g), ylo = fbbev_ld_bf16x8(yl + 32 * s + 8 * g);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const fbbev_bf16x8 ah = fbbev_ld_bf16x8(a2t + ((mt * KS + s) * 64 + lane) * 8);
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(a2t + A2 + ((mt * KS + s) * 64 + lane) * 8);
                acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(al, yho, acc2[mt]);
                acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(ah, ylo, acc2[mt]);
                acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(ah, yho, acc2[mt]);
            }
        }
        {
            fbbev_v4u* wd = reinterpret_cast<fbbev_v4u*>(a2buf + ((t + 1) & 1) * A2S);
#pragma unroll
            for (int q = 0; q < A2P; ++q) wd[threadIdx.x + 512 * q] = wst[q];
        }
    };
    for (int t = 0; t < T1; t += PF) {
        frame(t, fbbev_ic<0>{});
        if (t + 1 < T1) frame(t + 1, fbbev_ic<1>{});
        if (t + 2 < T1) frame(t + 2, fbbev_ic<2>{});
    }
    if (inb) {
        float* ob = out + (long long)b * Cout * N + n;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
    }
}

// Folded weights -> split bf16 A operands in fragment order: dst = [ w1 hi | w1 lo | per frame t: w2_t hi | w2_t lo ], a part =
// [mt][s][lane][8], element e of a lane = W[16 mt + lane % 16][32 s + 8 (lane / 16) + e] (zero beyond C).
__global__ void __launch_bounds__(256)
k_history_weight_fragments_bf16x3(const float* __restrict__ w1, const float* __restrict__ w2, int MT1, int MT2, int C, int T1,
                                  unsigned short* __restrict__ dst) {
    const int KS = (C + 31) / 32;
    const int n1 = MT1 * KS * 64, n2 = MT2 * KS * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                   // one lane-fragment (8 elements, both parts) per thread
    if (i >= n1 + T1 * n2) return;
    fbbev_v4f lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    const int ii = i < n1 ? i : (i - n1) % n2, t = i < n1 ? 0 : (i - n1) / n2;
    const int lane = ii & 63, s = (ii >> 6) % KS, mt = (ii >> 6) / KS;
    const float* row = i < n1 ? w1 + (long long)(16 * mt + (lane & 15)) * C
                              : w2 + (long long)(16 * mt + (lane & 15)) * ((long long)T1 * C) + (long long)t * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 32 * s + 8 * (lane >> 4) + e;
        const float v = c < C ? row[c] : 0.f;
        if (e < 4) lo[e] = v; else hi[e - 4] = v;
    }
    fbbev_bf16x8 h8, l8;
    fbbev_split_bf16x8(lo, hi, h8, l8);
    unsigned short* ph = i < n1 ? dst + (long long)ii * 8 : dst + 2ll * n1 * 8 + ((long long)t * 2 * n2 + ii) * 8;
    unsigned short* pl = ph + (long long)(i < n1 ? n1 : n2) * 8;
    __builtin_memcpy(ph, &h8, 16);
    __builtin_memcpy(pl, &l8, 16);
}
