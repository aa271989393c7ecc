// history_conv_x3_kernels.h -- the two folded 1x1x1 convolutions of the temporal fusion at fp32-GRADE precision on the 16-bit
// MFMAs (split operands; round 3, rebuilt in round 6).
//
// The exact route (k_history_conv_t: v_mfma_f32_16x16x4_f32) is compute bound: 10.5 ms of the 16.6 ms BASELINE configs[4] step.
// The bf16 route (k_history_conv_bf16) is 5x faster but rounds weights, frames and the intermediate to 8 mantissa bits (~4e-3
// of the output peak).  Here no operand loses more than ~2^-17 of itself:
//   * convolution 1 (x_t -> y_t, C x C): the ring element IS the MFMA operand -- an fp16 ring goes through
//     v_mfma_f32_16x16x32_f16 untouched, a bf16 ring through v_mfma_f32_16x16x32_bf16 -- and the folded weight is split into two
//     16-bit terms of that type, W = hi + lo (22 / 16 mantissa bits): TWO MFMAs per product, no conversion of the frames at all.
//     For the fp16 form a weight row is scaled by a power of two S_c so that its largest entry sits at 2^11 (hi and lo stay
//     normal halves whatever the layer's scale); the bias of that row is scaled with it, and column c of W2 by 1 / S_c --
//     all exact -- so the kernel never multiplies by a scale: y'_c = relu(acc + S_c b_c) = S_c y_c.
//   * convolution 2 (y_t -> out, Cout x T1 C): y' (fp32) and W2' are split into two bf16 terms each, a product is three MFMAs
//     a_hi.b_hi + a_lo.b_hi + a_hi.b_lo (the dropped lo.lo term and the split's own remainder are ~2^-17 relative).
// Error against the fp32 convolutions of the same stored frames: ~1e-5 of the output peak (tests) -- three decimal digits
// better than the TF32 arithmetic PyTorch's cuDNN convolutions use by default on the reference's own hardware.
//
// Round 6 layout (the round-3 kernel spent a frame as: 45 LDS fragment reads each waited for by the MFMA behind it,
// ~225 VALU operations converting frames and re-packing y through LDS, and 90 MFMAs per 16 voxels -- 8 850 cycles per frame per CU
// where the MFMAs need 2 900):
//   * a wave owns NV = 2 tiles of 16 voxels: an A fragment read from LDS feeds 2 x (2 or 3) MFMAs, and the fragments of one K step
//     (all output tiles, hi and lo) are requested together, ahead of the MFMAs that use them;
//   * y never goes through LDS: with the weights as the A operand a lane ends convolution 1 holding channels 16 mt + 4 (lane/16) +
//     r of voxel lane%16, and the K order of an MFMA is free as long as A and B agree -- W2's fragments are laid out so that K
//     slot (lane/16, e) of step s stands for channel 16 (2 s + e/4) + 4 (lane/16) + e%4: the B operand of step s is the pair of
//     accumulator tiles 2 s, 2 s + 1 of the same lane, split in registers;
//   * the frame's bias rides in the LDS block of W2_t (one staging stream, double buffered, one barrier per frame).
// Shape: voxel-major 16-bit ring only.  512 threads = 8 waves x 32 voxels per workgroup; LDS: W2 block 2 x 32 KB + W1 30 KB.
#pragma once
#include "rt.h"
#include "history_kernels.h"
#include "history_conv_kernels.h"

#include "x3_split.h"

#define FBBEV_HX3_NV 2                                    // 16-voxel tiles per wave

template <int ET>
__device__ __forceinline__ fbbev_v4f fbbev_mfma_16x16x32_raw(fbbev_v4u a, fbbev_v4u b, fbbev_v4f c) {
    if constexpr (ET == 2) return fbbev_mfma_f32_16x16x32_f16(a, b, c);
    else {
        fbbev_bf16x8 xa, xb;
        __builtin_memcpy(&xa, &a, 16);
        __builtin_memcpy(&xb, &b, 16);
        return fbbev_mfma_f32_16x16x32_bf16(xa, xb, c);
    }
}

template <int MT1, int MT2, int ET, int PF, int NW, int FPB>
__global__ void __launch_bounds__(64 * NW, 2)
k_history_conv_bf16x3(const void* __restrict__ feats, long long fstride_b, const unsigned short* __restrict__ w1x,
                      const float* __restrict__ biasx, const unsigned short* __restrict__ w2x, const float* __restrict__ bias2,
                      int T1, int N, int tiles_per_b, float* __restrict__ out, int seg0, int seg_stride, int seg_len,
                      int tiles_per_seg) {
    static_assert(ET == 1 || ET == 2, "16-bit voxel-major ring");
    constexpr int NV = FBBEV_HX3_NV, NT = 64 * NW;          // NW waves of NV x 16 voxels (8: one workgroup fills a CU's registers; 4: half of them)
    constexpr int C = 16 * MT1, Cout = 16 * MT2, KS = (C + 31) / 32;
    constexpr int A1 = MT1 * KS * 64 * 8;                 // 16-bit elements of one W1 part (hi or lo)
    constexpr int A2 = MT2 * KS * 64 * 8;                 // ... of one W2 part of one frame; a frame's block is [hi | lo | bias]
    constexpr int NW2 = 2 * A2 / 8, NBI = C / 4;          // 16-byte pieces of a frame's block: weights, then the C bias floats
    constexpr int NP = NW2 + NBI, A2P = (NP + NT - 1) / NT;
    constexpr int A2S = A2P * NT * 8;                     // elements of a staging buffer
    constexpr int NW1 = 2 * A1 / 8, IW1 = (NW1 + NT - 1) / NT;
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());
    // FPB frames per barrier: block t lives in buffer t % (2 FPB); the block staged during frame t is block t + FPB
    constexpr int NB = 2 * FPB;
    unsigned short* a2buf = lds;                          // [NB][A2S]
    unsigned short* w1buf = lds + NB * A2S;               // [hi A1 | lo A1]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    // the launch covers tiles_per_b / tiles_per_seg segments of seg_len voxels, seg_stride apart from seg0 on: the whole sample
    // (one segment of N voxels) or, for one chunk of the pipelined step, the rows of a y range in every z plane
    const int seg = tile / tiles_per_seg, ti = tile - seg * tiles_per_seg;
    int n[NV];
    bool inb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int nl = ti * (16 * NW * NV) + (wave * NV + v) * 16 + j;
        inb[v] = nl < seg_len;
        n[v] = seg0 + seg * seg_stride + nl;
    }
    // a frame's block: piece i < NW2 of W2_t, then the frame's (scaled) bias
    auto block_piece = [&](int t, int q) -> fbbev_v4u {
        const int i = (int)threadIdx.x + NT * q;
        const fbbev_v4u* wsrc = reinterpret_cast<const fbbev_v4u*>(w2x + (long long)t * 2 * A2);
        const fbbev_v4u* bsrc = reinterpret_cast<const fbbev_v4u*>(biasx + ((long long)b * T1 + t) * C);
        if (NT * (q + 1) <= NW2) return wsrc[i];                                      // the whole round is weights
        const fbbev_v4u* src = i < NW2 ? wsrc + i : bsrc + (i - NW2 < NBI ? i - NW2 : 0);
        return *src;
    };
    {   // block 0 and W1 (hi | lo): every piece REQUESTED before the first is stored
        fbbev_v4u ta[FPB][A2P], tb[IW1];
#pragma unroll
        for (int f = 0; f < FPB; ++f)
#pragma unroll
            for (int q = 0; q < A2P; ++q) ta[f][q] = block_piece(f < T1 ? f : T1 - 1, q);
#pragma unroll
        for (int k = 0; k < IW1; ++k) { const int i = (int)threadIdx.x + NT * k; tb[k] = reinterpret_cast<const fbbev_v4u*>(w1x)[i < NW1 ? i : 0]; }
#pragma unroll
        for (int f = 0; f < FPB; ++f)
#pragma unroll
            for (int q = 0; q < A2P; ++q) reinterpret_cast<fbbev_v4u*>(a2buf + f * A2S)[threadIdx.x + NT * q] = ta[f][q];
#pragma unroll
        for (int k = 0; k < IW1; ++k) { const int i = (int)threadIdx.x + NT * k; if (i < NW1) reinterpret_cast<fbbev_v4u*>(w1buf)[i] = tb[k]; }
    }
    const long long xb = (long long)b * fstride_b;
    fbbev_v4f acc2[NV][MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bz = bias2[16 * mt + 4 * g + r];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc2[v][mt][r] = bz;
        }
    fbbev_v4u bv[PF][NV][KS];                               // X stays RAW in registers: it is the B operand as loaded
    unsigned int xoff[NV][KS];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * g;
            xoff[v][s] = (inb[v] && c < C) ? (unsigned int)(((long long)n[v] * C + c) * 2) : 0u;
        }
    auto load_x = [&](int slot, long long base) {
        const char* fb = static_cast<const char*>(feats) + base * 2;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[slot][v][s] = *reinterpret_cast<const fbbev_v4u*>(fb + xoff[v][s]);
    };
    const long long fsz = (long long)C * N;
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (u < T1) load_x(u, xb + (long long)u * fsz);
    const fbbev_v4u zero4 = {0u, 0u, 0u, 0u};
    const fbbev_v4f zero4f = {0.f, 0.f, 0.f, 0.f};
    auto frame = [&](int t, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        if (t % FPB == 0) __syncthreads();                      // blocks t .. t + FPB - 1 are in place; the FPB buffers before them are free again
        const unsigned short* a2t = a2buf + (t % NB) * A2S;
        fbbev_v4f acc1[NV][MT1];                                // starts at the frame's bias: it is the C operand of the first MFMA
        {
            const float* bt = reinterpret_cast<const float*>(a2t + 2 * A2);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) {
                const fbbev_v4f bia = *reinterpret_cast<const fbbev_v4f*>(bt + 16 * mt + 4 * g);
#pragma unroll
                for (int v = 0; v < NV; ++v) acc1[v][mt] = bia;
            }
        }
        // ---- convolution 1: W1 (lo, hi) . x_t
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            fbbev_v4u af[MT1], xr[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                xr[v] = bv[SL][v][s];
                if (32 * s + 32 > C) xr[v] = (32 * s + 8 * g < C) ? xr[v] : zero4;      // K padding: 0 x (another voxel's bits) must stay 0
            }
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) af[mt] = *reinterpret_cast<const fbbev_v4u*>(w1buf + A1 + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) acc1[v][mt] = fbbev_mfma_16x16x32_raw<ET>(af[mt], xr[v], acc1[v][mt]);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) af[mt] = *reinterpret_cast<const fbbev_v4u*>(w1buf + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) acc1[v][mt] = fbbev_mfma_16x16x32_raw<ET>(af[mt], xr[v], acc1[v][mt]);
        }
        // issue order of the fragment reads and MFMAs above (the compiler on its own issues a read right in front of its MFMA and waits
        // for it): a fragment set (MT1 reads) is requested while the last MFMAs of the set before it issue -- one read behind each, into
        // the registers that MFMA just consumed.  The region's first MT1 LDS reads are the bias.
        if constexpr (NV == 2) {
            FBBEV_SCHED_LDS_READ(2 * MT1);
#pragma unroll
            for (int q = 0; q < 2 * KS; ++q) {
                FBBEV_SCHED_MFMA(MT1);                                                             // tile 0
                if (q + 1 < 2 * KS) {
#pragma unroll
                    for (int i = 0; i < MT1; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }  // tile 1: the set's last use
                } else FBBEV_SCHED_MFMA(MT1);
            }
        }
        fbbev_sched_fence();
        // the next block and the next frames of X: requested here, under the epilogue and convolution 2
        fbbev_v4u wst[A2P];
        {
            const int tn = t + FPB < T1 ? t + FPB : T1 - 1;
#pragma unroll
            for (int q = 0; q < A2P; ++q) wst[q] = block_piece(tn, q);
        }
        load_x(SL, xb + (long long)(t + PF < T1 ? t + PF : T1 - 1) * fsz);
        fbbev_sched_fence();
        // ---- y' = relu(acc) (the bias is in it), split into bf16 hi / lo in the K order of W2's fragments: step s = tiles 2 s, 2 s + 1
        fbbev_bf16x8 yh[NV][KS], yl[NV][KS];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                fbbev_v4f y0, y1 = zero4f;
#pragma unroll
                for (int r = 0; r < 4; ++r) y0[r] = fmaxf(acc1[v][2 * s][r], 0.f);
                if (2 * s + 1 < MT1) {
                    const int m1 = 2 * s + 1 < MT1 ? 2 * s + 1 : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y1[r] = fmaxf(acc1[v][m1][r], 0.f);
                }
                fbbev_split_bf16x8(y0, y1, yh[v][s], yl[v][s]);
            }
        // ---- convolution 2: acc2 += W2'_t . y'
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            fbbev_bf16x8 af[MT2];
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) af[mt] = fbbev_ld_bf16x8(a2t + A2 + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) acc2[v][mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yh[v][s], acc2[v][mt]);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) af[mt] = fbbev_ld_bf16x8(a2t + ((mt * KS + s) * 64 + lane) * 8);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) acc2[v][mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yl[v][s], acc2[v][mt]);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) acc2[v][mt] = fbbev_mfma_f32_16x16x32_bf16(af[mt], yh[v][s], acc2[v][mt]);
        }
        // issue order (as above): per K step the lo set feeds 2 x MT2 MFMAs, the hi set 4 x MT2
        if constexpr (NV == 2) {
            FBBEV_SCHED_LDS_READ(MT2);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                FBBEV_SCHED_MFMA(MT2);
#pragma unroll
                for (int i = 0; i < MT2; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }      // lo . yh, tile 1; the hi set behind it
                FBBEV_SCHED_MFMA(3 * MT2);
                if (s + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < MT2; ++i) { FBBEV_SCHED_MFMA(1); FBBEV_SCHED_LDS_READ(1); }  // hi . yh, tile 1; the next step's lo set
                } else FBBEV_SCHED_MFMA(MT2);
            }
        }
        {
            fbbev_v4u* wd = reinterpret_cast<fbbev_v4u*>(a2buf + ((t + FPB) % NB) * A2S);
#pragma unroll
            for (int q = 0; q < A2P; ++q) wd[threadIdx.x + NT * q] = wst[q];
        }
    };
    for (int t = 0; t < T1; t += PF) {
        frame(t, fbbev_ic<0>{});
        if constexpr (PF > 1) { if (t + 1 < T1) frame(t + 1, fbbev_ic<(PF > 1 ? 1 : 0)>{}); }
        if constexpr (PF > 2) { if (t + 2 < T1) frame(t + 2, fbbev_ic<(PF > 2 ? 2 : 0)>{}); }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (inb[v]) {
            float* ob = out + (long long)b * Cout * N + n[v];
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[v][mt][r], 0.f);
        }
}

// Row scale of convolution 1 (fp16 form): the power of two that puts the row's largest |w| in [2^11, 2^12) -- as float bits, with
// its reciprocal; 1 for a zero row or one outside 2^+-30 (nothing to gain there, and S y must stay far from the fp32 range's ends).
__device__ __forceinline__ void fbbev_hx3_row_scale(float rowmax, float& s, float& inv) {
    unsigned int u;
    __builtin_memcpy(&u, &rowmax, 4);
    const int eb = (int)((u >> 23) & 0xffu);
    unsigned int su = 0x3f800000u, iu = 0x3f800000u;
    if (eb >= 97 && eb <= 157) { su = (unsigned int)(265 - eb) << 23; iu = (unsigned int)(eb - 11) << 23; }
    __builtin_memcpy(&s, &su, 4);
    __builtin_memcpy(&inv, &iu, 4);
}

// Folded weights -> split 16-bit A operands in fragment order, and the biases of convolution 1 in its rows' scale:
//   dst  = [ w1 hi | w1 lo | per frame t: w2_t hi | w2_t lo ], a part = [mt][s][lane][8];
//   W1 (ET 2: halves of S_row w, ET 1: bf16 of w): element e of a lane = W1[16 mt + lane % 16][32 s + 8 (lane / 16) + e];
//   W2 (bf16 of w / S_c): element e = W2_t[16 mt + lane % 16][c], c = 16 (2 s + e / 4) + 4 (lane / 16) + e % 4 -- the channels a lane
//   of k_history_conv_bf16x3 holds after convolution 1 (zero beyond C);
//   bias_dst[bt][c] = S_c bias1[bt][c].
template <int ET>
__global__ void __launch_bounds__(256)
k_history_weight_fragments_bf16x3(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ bias1,
                                  int MT1, int MT2, int C, int T1, int BT, unsigned short* __restrict__ dst,
                                  float* __restrict__ bias_dst) {
    __shared__ float sc[256], isc[256];
    for (int c = threadIdx.x; c < C; c += 256) {
        float m = 0.f, s = 1.f, inv = 1.f;
        if constexpr (ET == 2) {
            for (int k = 0; k < C; ++k) m = fmaxf(m, fabsf(w1[(long long)c * C + k]));
            fbbev_hx3_row_scale(m, s, inv);
        }
        sc[c] = s; isc[c] = inv;
    }
    __syncthreads();
    const int KS = (C + 31) / 32;
    const int n1 = MT1 * KS * 64, n2 = MT2 * KS * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                   // one lane-fragment (8 elements, both parts) per thread
    if (i >= n1 + T1 * n2) {
        const int k = i - (n1 + T1 * n2);                                  // then one bias per thread
        if (k < BT * C) bias_dst[k] = bias1[k] * sc[k % C];
        return;
    }
    fbbev_v4f lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    const int ii = i < n1 ? i : (i - n1) % n2, t = i < n1 ? 0 : (i - n1) / n2;
    const int lane = ii & 63, s = (ii >> 6) % KS, mt = (ii >> 6) / KS;
    const int row = 16 * mt + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        if (i < n1) {
            const int c = 32 * s + 8 * g + e;
            if (c < C) v = w1[(long long)row * C + c] * sc[row];
        } else {
            const int c = 16 * (2 * s + (e >> 2)) + 4 * g + (e & 3);
            if (c < C) v = w2[(long long)row * ((long long)T1 * C) + (long long)t * C + c] * isc[c];
        }
        if (e < 4) lo[e] = v; else hi[e - 4] = v;
    }
    unsigned short* ph = i < n1 ? dst + (long long)ii * 8 : dst + 2ll * n1 * 8 + ((long long)t * 2 * n2 + ii) * 8;
    unsigned short* pl = ph + (long long)(i < n1 ? n1 : n2) * 8;
    if (ET == 2 && i < n1) {                                               // two halves: hi = f16(v), lo = f16(v - hi)
        fbbev_v4u h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = e < 2 ? lo[2 * e] : hi[2 * e - 4], c = e < 2 ? lo[2 * e + 1] : hi[2 * e - 3];
            h[e] = fbbev_cvt_pk16<2>(a, c);
            const float ra = a - fbbev_f16_bits_to_f32(h[e] & 0xffffu), rc = c - fbbev_f16_bits_to_f32(h[e] >> 16);
            l[e] = fbbev_cvt_pk16<2>(ra, rc);
        }
        __builtin_memcpy(ph, &h, 16);
        __builtin_memcpy(pl, &l, 16);
        return;
    }
    fbbev_bf16x8 h8, l8;
    fbbev_split_bf16x8(lo, hi, h8, l8);
    __builtin_memcpy(ph, &h8, 16);
    __builtin_memcpy(pl, &l8, 16);
}
