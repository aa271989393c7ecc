// history_conv_kernels.h -- the two 1x1x1 convolutions of FB-OCC's temporal fusion as ONE fp32-MFMA kernel.
//
// Replaces, for inference, history_keyframe_time_conv + history_keyframe_cat_conv of FBOCC.fuse_history
// (mmdet3d/models/fbbev/detectors/fbocc.py:111-127, 289-310): the reference concatenates a time channel (81 channels),
// runs Conv3d(81->80)+BN+ReLU on each of the T+1 frames, concatenates the results (1360 channels) and runs
// Conv3d(1360->80)+BN+ReLU -- the (T+1)*C-channel intermediate is written and read back (435 MB per sample at
// 100x100x8).  With the eval-mode batch norms folded into the weights and the time channel into a per-frame bias
// (history_fusion.py) the math per voxel n is
//     out[:, n] = relu( b2 + sum_t  W2_t . relu( W1 . x_t[:, n] + b1_t ) ),      W1: CxC, W2_t: Cout x C
// which is GEMM-shaped (2 * 2 * C * C * (T+1) flops per voxel = 35 GFLOP per 100x100x8 sample) and therefore belongs
// on the matrix cores.  The path stays fp32 (the reference pins fuse_history to fp32, fbocc.py:207 @force_fp32), so
// this uses v_mfma_f32_16x16x4_f32: exact f32, bit-for-bit a k-ordered fmaf chain (MI355X guide: 157 TF peak).
//
// Tiling (wave64): a workgroup of 4 waves owns 64 consecutive voxels, wave w the 16-voxel N-tile n0+16w.  Per frame t
// the wave computes Y_t = relu(W1 . X_t + b1_t) for ALL C output channels (C/16 M-tiles x C/4 k-steps; the B fragment
// of k-step kk is X_t[4kk + lane/16][n0 + lane%16], read straight from HBM: 4 rows x 64 contiguous bytes), parks Y_t in a
// wave-private LDS slab in the MFMA C layout, reads it back in the B layout and accumulates Out += W2_t . Y_t.
// The intermediate never leaves the CU.  Fragment layouts of v_mfma_f32_16x16x4_f32 (A 16x4, B 4x16, C/D 16x16):
//     A: lane holds A[i = lane%16][k = lane/16]      B: lane holds B[k = lane/16][j = lane%16]
//     D: register r of a lane holds D[i = 4*(lane/16) + r][j = lane%16]
// Bound: fp32 MFMA (64 FLOP/clk/SIMD); HBM traffic = the (T+1)*C-channel volume read once + the output written once.
#pragma once
#include "rt.h"
#include "history_kernels.h"    // fbbev_ld_elem: 16-bit storage of the frame buffer (widened exactly, fp32 MFMA as before)

#define FBBEV_HC_MAX_TILES 8      // C, Cout <= 128 (M-tiles of 16)

// feats (B, T1*C, N) with batch stride fstride_b; w1 (C,C); bias1 (B*T1, C); w2 (Cout, T1*C); bias2 (Cout); out (B,Cout,N)
template <int ET>
__global__ void __launch_bounds__(256)
k_history_conv(const void* __restrict__ feats, long long fstride_b, const float* __restrict__ w1,
               const float* __restrict__ bias1, const float* __restrict__ w2, const float* __restrict__ bias2,
               int T1, int C, int Cout, int N, int tiles_per_b, float* __restrict__ out) {
    float* lds = fbbev_dyn_lds_f32();                      // [4 waves][C][16]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int n = tile * 64 + wave * 16 + j;
    const bool inb = n < N;
    const int MT1 = C >> 4, MT2 = Cout >> 4, KS = C >> 2;
    float* ylds = lds + wave * C * 16;
    const long long xb = (long long)b * fstride_b;        // element offset of this sample's frames
    fbbev_v4f acc2[FBBEV_HC_MAX_TILES];
#pragma unroll
    for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt) {
        acc2[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        if (mt < MT2)
            for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
    }
    for (int t = 0; t < T1; ++t) {
        const long long xt = xb + (long long)t * C * N;
        const float* b1 = bias1 + ((long long)b * T1 + t) * C;
        fbbev_v4f acc1[FBBEV_HC_MAX_TILES];
#pragma unroll
        for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt) {
            acc1[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            if (mt < MT1)
                for (int r = 0; r < 4; ++r) acc1[mt][r] = b1[16 * mt + 4 * g + r];
        }
        for (int kk = 0; kk < KS; ++kk) {
            const float bf = inb ? fbbev_ld_elem<ET>(feats, xt + (long long)(4 * kk + g) * N + n) : 0.f;
#pragma unroll
            for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt)
                if (mt < MT1) acc1[mt] = fbbev_mfma_f32_16x16x4(w1[(16 * mt + j) * C + 4 * kk + g], bf, acc1[mt]);
        }
        __syncthreads();                                    // the previous frame's reads of ylds are done
#pragma unroll
        for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt)
            if (mt < MT1)
                for (int r = 0; r < 4; ++r) ylds[(16 * mt + 4 * g + r) * 16 + j] = fmaxf(acc1[mt][r], 0.f);
        __syncthreads();
        const float* w2t = w2 + (long long)t * C;
        const long long w2ld = (long long)T1 * C;
        for (int kk = 0; kk < KS; ++kk) {
            const float bf = ylds[(4 * kk + g) * 16 + j];
#pragma unroll
            for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt)
                if (mt < MT2) acc2[mt] = fbbev_mfma_f32_16x16x4(w2t[(16 * mt + j) * w2ld + 4 * kk + g], bf, acc2[mt]);
        }
    }
    if (inb) {
        float* ob = out + (long long)b * Cout * N + n;
#pragma unroll
        for (int mt = 0; mt < FBBEV_HC_MAX_TILES; ++mt)
            if (mt < MT2)
                for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
    }
}


// Register-resident variant for compile-time channel counts (C = 16*MT1, Cout = 16*MT2).  In the generic kernel above
// every MFMA waits for its own scattered dword load of a weight fragment (measured 12x off the MFMA bound).  Here
//   * the weights arrive PRE-ARRANGED in fragment order (w1f[mt][kk][lane], w2f[t][mt][kk][lane]: the host permutes the
//     folded weight matrices once), so a fragment load is one coalesced 256-byte wave load;
//   * the W1 fragments live in registers for the whole kernel, the W2_t fragments of a frame are loaded in one burst;
//   * the X fragments of frame t+1 are fetched while frame t's second GEMM runs (register double buffer).
// (Streaming the W2 fragments instead of the per-frame burst, to fit two waves per SIMD, measured 1.9x SLOWER on the
// same box -- 1.77 vs 0.94 ms for the whole fusion step -- and was dropped.)
// VM: voxel-major frames ([T1][N][C] per sample, history_kernels.h).  K slot kk of lane group g then stands for channel
// 16 (kk / 4) + 4 g + kk % 4 in GEMM 1 (k_history_weight_fragments arranges W1 the same way): a lane's four channels of a
// 16-channel block are one 8-byte (16-bit ring) / 16-byte (fp32 ring) row piece.  The fp32 products are the same, their
// order inside the accumulation chain is permuted: equal to the planar kernel to fp32 rounding, not bit for bit.
template <int MT1, int MT2, int ET, bool VM = false>
__global__ void __launch_bounds__(256)
k_history_conv_t(const void* __restrict__ feats, long long fstride_b, const float* __restrict__ w1f,
                 const float* __restrict__ bias1, const float* __restrict__ w2f, const float* __restrict__ bias2,
                 int T1, int N, int tiles_per_b, float* __restrict__ out) {
    constexpr int C = 16 * MT1, Cout = 16 * MT2, KS = C / 4;
    float* lds = fbbev_dyn_lds_f32();                      // [4 waves][C][16]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int n = tile * 64 + wave * 16 + j;
    const bool inb = n < N;
    float* ylds = lds + wave * C * 16;
    const long long xb = (long long)b * fstride_b;        // element offset of this sample's frames
    float a1[MT1][KS];
#pragma unroll
    for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a1[mt][kk] = w1f[(mt * KS + kk) * 64 + lane];
    fbbev_v4f acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
    // X fragments stay RAW (unconverted) in registers: the prefetch of frame t+1 is issued before GEMM 2 of frame t and
    // must not be followed by a conversion that waits for it; the exact widening happens when the MFMA consumes them
    unsigned int bx[KS];
    constexpr int ESZ = ET == 0 ? 4 : 2;
    constexpr int PW = ET == 0 ? 4 : 2;                      // dwords of a 4-channel row piece
    unsigned int xoff[MT1];                                  // VM: lane byte offset of the piece of 16-channel block q (0 when out of range)
#pragma unroll
    for (int q = 0; q < MT1; ++q) xoff[q] = inb ? (unsigned int)(((long long)n * C + 16 * q + 4 * g) * ESZ) : 0u;
    auto load_x = [&](long long base) {                      // base: element offset of the frame
        if constexpr (VM) {
            const char* fb = static_cast<const char*>(feats) + base * ESZ;
#pragma unroll
            for (int q = 0; q < MT1; ++q) {                  // unconditional loads; the zero of an out-of-range voxel is selected at use
                const unsigned int* p = reinterpret_cast<const unsigned int*>(fb + xoff[q]);
                if constexpr (ET == 0) {
                    const fbbev_v4u v = *reinterpret_cast<const fbbev_v4u*>(p);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bx[4 * q + e] = v[e];
                } else {
                    const unsigned long long v = *reinterpret_cast<const unsigned long long*>(p);
                    bx[4 * q] = (unsigned int)v;             // channels 0,1 of the piece; [4q+1] holds 2,3; the other two slots are unused
                    bx[4 * q + 1] = (unsigned int)(v >> 32);
                }
            }
        } else {
            const int nc = inb ? n : 0;                      // clamped lane offset: unconditional loads, zero selected at use
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) bx[kk] = fbbev_ld_raw<ET>(feats, base + (long long)(4 * kk + g) * N + nc);
        }
    };
    auto x_at = [&](int kk) {                                // exact widening at the point of use
        if constexpr (!VM) return inb ? fbbev_widen<ET>(bx[kk]) : 0.f;
        else if constexpr (ET == 0) return inb ? fbbev_widen<0>(bx[kk]) : 0.f;
        else {
            const unsigned int w = bx[4 * (kk >> 2) + ((kk & 3) >> 1)];
            return inb ? fbbev_widen<ET>((kk & 1) ? (w >> 16) : (w & 0xffffu)) : 0.f;
        }
    };
    (void)PW;
    load_x(xb);
    for (int t = 0; t < T1; ++t) {
        const float* b1 = bias1 + ((long long)b * T1 + t) * C;
        const float* w2t = w2f + (long long)t * MT2 * KS * 64;
        // the frame's bias FIRST, the W2_t burst behind it: vmcnt retires in order, so with the bias loads behind the 100
        // fragment loads GEMM 1 waited for the whole burst every frame (one wave per SIMD: nothing else to run meanwhile)
        fbbev_v4f acc1[MT1];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) acc1[mt] = *reinterpret_cast<const fbbev_v4f*>(b1 + 16 * mt + 4 * g);
        fbbev_sched_fence();
        float a2[MT2][KS];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a2[mt][kk] = w2t[(mt * KS + kk) * 64 + lane];
        fbbev_sched_fence();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc1[mt] = fbbev_mfma_f32_16x16x4(a1[mt][kk], x_at(kk), acc1[mt]);
        if constexpr (VM) load_x(xb + (long long)(t + 1 < T1 ? t + 1 : t) * C * N);      // unconditional (clamped at the tail)
        else if (t + 1 < T1) load_x(xb + (long long)(t + 1) * C * N);                    // next frame's X fragments: in flight during GEMM 2
        fbbev_wave_sync();                                  // ylds is wave-private: no workgroup barrier
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ylds[(16 * mt + 4 * g + r) * 16 + j] = fmaxf(acc1[mt][r], 0.f);
        fbbev_wave_sync();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const float by = ylds[(4 * kk + g) * 16 + j];
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) acc2[mt] = fbbev_mfma_f32_16x16x4(a2[mt][kk], by, acc2[mt]);
        }
    }
    if (inb) {
        float* ob = out + (long long)b * Cout * N + n;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
    }
}

// Weight matrices -> MFMA A-fragment order, one launch: dst = [ w1f[mt][kk][lane] | w2f[t][mt][kk][lane] ] with
// fragment element A[16mt + lane%16][4kk + lane/16]; w1 is (16*MT1, C), w2 is (16*MT2, T1*C) and frame t uses columns t*C..
// vm: W1's K slots follow the voxel-major channel rule of k_history_conv_t<.., VM = true>.
__global__ void __launch_bounds__(256)
k_history_weight_fragments(const float* __restrict__ w1, const float* __restrict__ w2, int MT1, int MT2, int KS, int T1,
                           int vm, float* __restrict__ dst) {
    const int n1 = MT1 * KS * 64, n2 = MT2 * KS * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1 + T1 * n2) return;
    const int C = 4 * KS;
    if (i < n1) {
        const int lane = i & 63, kk = (i >> 6) % KS, mt = (i >> 6) / KS;
        const int col = vm ? 16 * (kk >> 2) + 4 * (lane >> 4) + (kk & 3) : 4 * kk + (lane >> 4);
        dst[i] = w1[(long long)(16 * mt + (lane & 15)) * C + col];
    } else {
        const int r = i - n1, t = r / n2, e = r - t * n2;
        const int lane = e & 63, kk = (e >> 6) % KS, mt = (e >> 6) / KS;
        dst[i] = w2[(long long)(16 * mt + (lane & 15)) * ((long long)T1 * C) + t * C + 4 * kk + (lane >> 4)];
    }
}


// ---------------------------------------------------------------- bf16-MFMA variant (opt-in: compute = bf16)
// BASELINE configs[4] names fp16 for the 16-frame temporal path; the reference pins these two convolutions to fp32
// (fbocc.py:279-282 force_fp32).  With the ring stored in 16 bits the fp32-MFMA kernel above is COMPUTE bound: 13-14 ms of
// the 23 ms frame at 400x400x16 (1.1 TFLOP at ~0.5 of the 157 TFLOP/s fp32-MFMA peak) against 1.2 ms to read the 6.1 GB
// ring.  This variant runs both GEMMs on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): operands rounded to bf16 -- the
// folded weights once on the host side of the launch (k_history_weight_fragments_bf16), the frames as stored (bf16 ring:
// exact; fp32 / fp16 ring: rounded at use), the ReLU'd intermediate when it is parked in LDS.  Biases, accumulators and
// the output stay fp32.  Channels are padded to a multiple of 32 with zeros (C = 80 -> 3 k-steps).
// Operand slots: lane (j = lane % 16, g = lane / 16), element e of k-step s stands for channel 32 s + 8 g + e in A and B
// alike (the instruction only needs the two to agree).
template <int V> struct fbbev_ic { static constexpr int value = V; };

// VM: the frames are voxel-major ([T1][N][C] per sample, history_kernels.h): a lane's 8 channels of a k-step are one 16-byte
// row piece (two for an fp32 ring), and for a bf16 ring that piece IS the MFMA operand; three frames of X are kept in
// flight (12 registers each in 16 bits) -- with 8 waves per CU one frame ahead is 20 KB per CU in flight, 2.5 TB/s at
// HBM latency.
// W2_t (15 KB of fragments per frame at C = 80) is the same for the four waves of a workgroup: it is staged through LDS
// (double-buffered, one barrier per frame: global -> registers before GEMM 1, registers -> LDS after GEMM 2) instead of
// every wave pulling its own copy through the vector L1 -- 6x the bytes of X itself.
template <int MT1, int MT2, int ET, bool VM>
__global__ void __launch_bounds__(256, 2)
k_history_conv_bf16(const void* __restrict__ feats, long long fstride_b, const unsigned short* __restrict__ w1f,
                    const float* __restrict__ bias1, const unsigned short* __restrict__ w2f, const float* __restrict__ bias2,
                    int T1, int N, int tiles_per_b, float* __restrict__ out) {
    constexpr int C = 16 * MT1, Cout = 16 * MT2, KS = (C + 31) / 32, CP = KS * 32, PITCH = CP + 8;   // 16-byte aligned rows
    constexpr int A2 = MT2 * KS * 64 * 8;                 // bf16 elements of one frame's W2 fragments
    constexpr int A2P = (A2 / 8 + 255) / 256;             // 16-byte pieces of them per thread
    constexpr int A2S = A2P * 256 * 8;                    // elements of a staging buffer (padded: every thread stores A2P pieces)
    constexpr int PF = VM ? 3 : 1;                        // frames of X in flight
    constexpr int NV = (VM && ET == 0) ? 2 : 1;
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());
    unsigned short* a2buf = lds;                          // [2][A2S]; then [4 waves][16 voxels][PITCH] bf16
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int n = tile * 64 + wave * 16 + j;
    const bool inb = n < N;
    unsigned short* yrow = lds + 2 * A2S + (wave * 16 + j) * PITCH;
    for (int c = C + g; c < CP; c += 4) yrow[c] = 0;                       // padding channels of the intermediate
    for (int i = threadIdx.x; i < A2 / 8; i += 256)                         // W2_0
        reinterpret_cast<fbbev_v4u*>(a2buf)[i] = reinterpret_cast<const fbbev_v4u*>(w2f)[i];
    const long long xb = (long long)b * fstride_b;
    fbbev_bf16x8 a1[MT1][KS];
#pragma unroll
    for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int s = 0; s < KS; ++s) a1[mt][s] = fbbev_ld_bf16x8(w1f + ((mt * KS + s) * 64 + lane) * 8);
    fbbev_v4f acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
    // X operands stay RAW in registers (a prefetch must not be followed by a conversion that waits for it)
    unsigned int bx[VM ? 1 : KS][VM ? 1 : 8];
    fbbev_v4u bv[VM ? PF : 1][VM ? KS : 1][NV];
    // unconditional loads from uniform frame base + a 32-bit lane byte offset that is 0 for an out-of-range voxel / padding
    // piece (selected once, as an offset: a select between addresses becomes a branch around the load); the zero itself is
    // selected at the point of use -- a select right behind the load would make the prefetch wait for its own data
    constexpr int ESZ = ET == 0 ? 4 : 2;
    unsigned int xoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        xoff[s] = (inb && c < C) ? (unsigned int)(((long long)n * C + c) * ESZ) : 0u;      // C % 8 == 0: a piece is all in or all out
    }
    auto load_x = [&](int slot, long long base) {
        if constexpr (VM) {
            const char* fb = static_cast<const char*>(feats) + base * ESZ;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const fbbev_v4u* p4 = reinterpret_cast<const fbbev_v4u*>(fb + xoff[s]);
                bv[slot][s][0] = p4[0];
                if constexpr (ET == 0) bv[slot][s][NV - 1] = p4[1];
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 32 * s + 8 * g + e;
                    bx[s][e] = (inb && c < C) ? fbbev_ld_raw<ET>(feats, base + (long long)c * N + n) : 0u;
                }
        }
    };
    auto x_operand = [&](int sl, int s) {                       // sl: the slot frame t lives in (compile-time after unrolling)
        const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
        const bool ok = inb && 32 * s + 8 * g < C;
        if constexpr (VM && ET == 1) {
            fbbev_bf16x8 r;
            __builtin_memcpy(&r, &bv[sl][s][0], 16);             // a bf16 ring row piece is the operand
            return ok ? r : zero8;
        } else {
            fbbev_v4f lo, hi;
            if constexpr (VM && ET == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = fbbev_widen<0>(bv[sl][s][0][e]); hi[e] = fbbev_widen<0>(bv[sl][s][NV - 1][e]); }
            } else if constexpr (VM) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    lo[2 * e] = fbbev_widen<ET>(bv[sl][s][0][e] & 0xffffu);     lo[2 * e + 1] = fbbev_widen<ET>(bv[sl][s][0][e] >> 16);
                    hi[2 * e] = fbbev_widen<ET>(bv[sl][s][0][2 + e] & 0xffffu); hi[2 * e + 1] = fbbev_widen<ET>(bv[sl][s][0][2 + e] >> 16);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = fbbev_widen<ET>(bx[s][e]); hi[e] = fbbev_widen<ET>(bx[s][4 + e]); }
            }
            const fbbev_bf16x8 r = fbbev_cvt_bf16x8(lo, hi);    // exact for a bf16 ring
            if constexpr (VM) return ok ? r : zero8;
            else return r;
        }
    };
    const long long fsz = (long long)C * N;                     // elements of a frame (either layout)
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (u < T1) load_x(u, xb + (long long)u * fsz);
    // one frame; SL = the register slot its X lives in.  The slots are a ring indexed at compile time (the frame loop is
    // unrolled PF times): moving in-flight registers from slot to slot would wait for their loads.
    auto frame = [&](int t, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        __syncthreads();                                        // W2_t is in a2buf[t & 1]; a2buf[(t + 1) & 1] is free again
        const float* b1 = bias1 + ((long long)b * T1 + t) * C;
        // loads below are unconditional (clamped frame index at the tail): a load under a branch makes the wait counts after
        // the join assume it was NOT issued, i.e. wait for everything
        fbbev_v4u wst[A2P];
        {
            const fbbev_v4u* wn = reinterpret_cast<const fbbev_v4u*>(w2f + (long long)(t + 1 < T1 ? t + 1 : t) * A2);
#pragma unroll
            for (int q = 0; q < A2P; ++q) {
                const int i = threadIdx.x + 256 * q;
                wst[q] = wn[i < A2 / 8 ? i : 0];
            }
        }
        fbbev_sched_fence();
        // bias_1 of the frame: loaded now, added to the finished accumulator (its latency hides behind GEMM 1)
        fbbev_v4f bia[MT1], acc1[MT1];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            bia[mt] = *reinterpret_cast<const fbbev_v4f*>(b1 + 16 * mt + 4 * g);
            acc1[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        }
        fbbev_sched_fence();                                    // ... and issued BEFORE the X prefetch below (vmcnt is in order)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const fbbev_bf16x8 xo = x_operand(SL, s);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc1[mt] = fbbev_mfma_f32_16x16x32_bf16(a1[mt][s], xo, acc1[mt]);
        }
        fbbev_sched_fence();
        load_x(SL, xb + (long long)(t + PF < T1 ? t + PF : T1 - 1) * fsz);  // the slot just consumed takes frame t + PF
        fbbev_sched_fence();
        fbbev_wave_sync();                                                  // the Y rows are wave-private
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            // rows 16 mt + 4 g + r of voxel j = 4 consecutive channels of this lane's voxel row: one 8-byte store
            const fbbev_v4f y = {fmaxf(acc1[mt][0] + bia[mt][0], 0.f), fmaxf(acc1[mt][1] + bia[mt][1], 0.f),
                                 fmaxf(acc1[mt][2] + bia[mt][2], 0.f), fmaxf(acc1[mt][3] + bia[mt][3], 0.f)};
            const fbbev_bf16x8 pk = fbbev_cvt_bf16x8(y, y);
            unsigned long long four;
            __builtin_memcpy(&four, &pk, 8);
            *reinterpret_cast<unsigned long long*>(yrow + 16 * mt + 4 * g) = four;
        }
        fbbev_wave_sync();
        const unsigned short* a2t = a2buf + (t & 1) * A2S;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const fbbev_bf16x8 yo = fbbev_ld_bf16x8(yrow + 32 * s + 8 * g);   // channels 32 s + 8 g .. + 7 of voxel j
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
                acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(fbbev_ld_bf16x8(a2t + ((mt * KS + s) * 64 + lane) * 8), yo, acc2[mt]);
        }
        {   // unconditional (a use under a branch lets the compiler sink the loads down to it): the last frame stores a copy
            // of its own W2 into the idle buffer, the padding pieces take piece 0
            fbbev_v4u* wd = reinterpret_cast<fbbev_v4u*>(a2buf + ((t + 1) & 1) * A2S);
#pragma unroll
            for (int q = 0; q < A2P; ++q) wd[threadIdx.x + 256 * q] = wst[q];
        }
    };
    for (int t = 0; t < T1; t += PF) {
        frame(t, fbbev_ic<0>{});
        if constexpr (PF > 1) { if (t + 1 < T1) frame(t + 1, fbbev_ic<(PF > 1 ? 1 : 0)>{}); }
        if constexpr (PF > 2) { if (t + 2 < T1) frame(t + 2, fbbev_ic<(PF > 2 ? 2 : 0)>{}); }
    }
    if (inb) {
        float* ob = out + (long long)b * Cout * N + n;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
    }
}

// Folded weight matrices -> bf16 A operands in fragment order: dst = [ w1f[mt][s][lane][8] | w2f[t][mt][s][lane][8] ],
// element e of lane = W[16 mt + lane % 16][32 s + 8 (lane / 16) + e] (zero beyond C), rounded to nearest even.
__global__ void __launch_bounds__(256)
k_history_weight_fragments_bf16(const float* __restrict__ w1, const float* __restrict__ w2, int MT1, int MT2, int C, int T1,
                                unsigned short* __restrict__ dst) {
    const int KS = (C + 31) / 32;
    const int n1 = MT1 * KS * 64, n2 = MT2 * KS * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                   // one lane-fragment (8 elements) per thread
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
    const fbbev_bf16x8 pk = fbbev_cvt_bf16x8(lo, hi);
    __builtin_memcpy(dst + (long long)i * 8, &pk, 16);
}
