// history_fused_kernels.h -- temporal history fusion as ONE kernel: trilinear warp of the T history frames, the new ring,
// and the two folded 1x1x1 convolutions (round 3; SURVEY 8f-1: "a fused warp-and-1x1x1-conv HIP kernel").
//
// The two-kernel form (k_history_warp_vm, then k_history_conv_bf16) writes the T warped frames to the next ring and reads
// them straight back as the convolutions' operands: 6.55 GB written + 6.55 GB re-read per frame at 400x400x16 (fp16 ring).
// Here a workgroup owns 64 voxels of one grid row for all frames and splits into two ROLES:
//   * 12 producer waves: thread = one (voxel, 8-channel group) item of the tile; per frame the 8 trilinear taps (16-byte loads of
//     the PREVIOUS ring, the taps / weights of fbbev_warp_taps_vm set up once per item -- the flow is the sample's), the
//     blend in k_history_warp_vm's order (=> the SAME stored element bits), one rounding to the ring's element type, the
//     16-byte row piece stored to the NEXT ring (non-temporal) AND, as the bf16 MFMA operand, into an LDS tile [64][88];
//   * 4 consumer waves: k_history_conv_bf16's frame body -- GEMM 1 (W1 fragments register-resident), bias + ReLU, the
//     wave-private Y rows, GEMM 2 against the LDS-staged W2_t -- with the X operand read from the LDS tile instead of the
//     ring.  Operand bits equal the two-kernel path's: a bf16 ring piece is the operand, an fp16 piece is widened and rounded
//     to bf16 exactly as k_history_conv_bf16 does when it loads it.
// One workgroup barrier per frame (the one the W2 double buffer already needs): producers fill tile (t+1) & 1 while the
// consumers work on tile t & 1.  Why roles and not "every MFMA wave warps its own operands": with vmcnt retiring in issue order
// a wave that has 24 taps in flight cannot get at its bias / W2 loads, and 96 more registers per frame in flight do not fit
// next to the weight fragments; and why 8 producer waves: profiles/r03_exp_history_occupancy.jsonl -- the warp keeps 95 % of
// its speed at 8 waves per CU (3.68 vs 3.49 ms) but not at 4 (4.77 ms), the convolutions lose 0.7 ms at 4 waves per CU.
// C = Cout = 80, 16-bit voxel-major rings (BASELINE configs[4] names fp16); other shapes take the two-kernel path.
#pragma once
#include "rt.h"
#include "history_kernels.h"
#include "history_conv_kernels.h"

#define FBBEV_HF_XP 88             // pitch (elements) of an X-tile row: 80 channels + 8 (ds_read_b128 of 16 voxels: conflict-free)

// bf16 MFMA operand bits of a 16-byte ring piece (8 elements of type ET)
template <int ET>
__device__ __forceinline__ fbbev_v4u fbbev_hf_operand(fbbev_v4u piece) {
    if constexpr (ET == 1) return piece;
    else {
        fbbev_v4f lo, hi;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            lo[2 * e] = fbbev_widen<ET>(piece[e] & 0xffffu);     lo[2 * e + 1] = fbbev_widen<ET>(piece[e] >> 16);
            hi[2 * e] = fbbev_widen<ET>(piece[2 + e] & 0xffffu); hi[2 * e + 1] = fbbev_widen<ET>(piece[2 + e] >> 16);
        }
        const fbbev_bf16x8 r = fbbev_cvt_bf16x8(lo, hi);
        fbbev_v4u u;
        __builtin_memcpy(&u, &r, 16);
        return u;
    }
}

// Version 2 (the first build -- 8 producer waves, one frame per barrier, W1 fragments in registers -- measured 7.65 ms against
// 5.56 ms for the two kernels at 400x400x16: with 166 registers only ONE workgroup fits a CU, and with one frame per barrier a
// CU never had more than that frame's 82 KB of taps in flight, half of what the stand-alone warp kernel keeps up).  Now:
//   * FRAME PAIRS: producers issue the taps of two frames (16 loads per thread) before they blend, consumers run two frames
//     per barrier; four operand tiles (2 pairs) and four W2 buffers in LDS;
//   * the W1 fragments live in LDS (15 KB, read conflict-free: consecutive lanes, consecutive 16-byte pieces) instead of 60
//     registers: <= 128 registers, so a 1024-thread workgroup (4 consumer + 12 producer waves = 4 waves per SIMD) fits, one item
//     (voxel, 8 channels) per producer thread.
template <int ET>
__global__ void __launch_bounds__(1024)
k_history_fused_bf16(const void* __restrict__ hist, long long hist_stride_b, void* __restrict__ nxt, long long nxt_stride_b,
                     const float* __restrict__ flow, const unsigned short* __restrict__ w1f, const float* __restrict__ bias1,
                     const unsigned short* __restrict__ w2f, const float* __restrict__ bias2, int T1, int Z, int Y, int X,
                     int n_xc, int YB, int nyb, int per_xcd, int n_work, float* __restrict__ out) {
    static_assert(ET == 1 || ET == 2, "16-bit rings");
    constexpr int MT1 = 5, MT2 = 5, C = 80, Cout = 80, KS = 3, CP = KS * 32, PITCH = CP + 8, XP = FBBEV_HF_XP;
    constexpr int A1 = MT1 * KS * 64 * 8;                 // bf16 elements of the W1 fragments
    constexpr int A2 = MT2 * KS * 64 * 8;                 // bf16 elements of one frame's W2 fragments
    constexpr int A2P = (A2 / 8 + 255) / 256;             // 16-byte pieces of them per consumer thread
    constexpr int A2S = A2P * 256 * 8;
    constexpr int ITEMS = 64 * (C / 8);                   // (voxel, channel group) items of a tile: one per producer thread
    constexpr int TILE = 64 * XP;
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());
    unsigned short* a2buf = lds;                          // [4][A2S]: W2 of frames f at slot f & 3
    unsigned short* a1lds = lds + 4 * A2S;                // [A1]
    unsigned short* ybase = a1lds + A1;                   // [4 waves][16 voxels][PITCH]
    unsigned short* xt = ybase + 4 * 16 * PITCH;          // [4][64 voxels][XP]: operand tile of frame f at slot f & 3
    float* b1lds = reinterpret_cast<float*>(xt + 4 * TILE);  // [4][C]: conv-1 bias (time channel folded in) of frame f at slot f & 3
    int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);            // one contiguous eighth per XCD (slab order)
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= n_work) return;
    const int z = work % Z; work /= Z;
    const int yl = work % YB; work /= YB;
    const int xc = work % n_xc; work /= n_xc;
    const int yb = work % nyb, b = work / nyb;
    const int y = yb * YB + yl;
    if (y >= Y) return;                                   // block-uniform
    const int N = Z * Y * X;
    const long long row0 = ((long long)z * Y + y) * X + (long long)xc * 64;          // voxel index of the tile's first voxel
    const size_t frame_bytes = (size_t)N * C * 2;
    const int steps = (T1 + 1) / 2;                       // frame pairs
    const bool consumer = threadIdx.x < 256;

    if (consumer) {
        // ------------------------------------------------------------------ the convolutions (k_history_conv_bf16's frame body)
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int g = lane >> 4, j = lane & 15;
        const int vl = wave * 16 + j;
        const bool inb = xc * 64 + vl < X;
        const long long n = row0 + vl;
        unsigned short* yrow = ybase + (wave * 16 + j) * PITCH;
        for (int c = C + g; c < CP; c += 4) yrow[c] = 0;                       // padding channels of the intermediate
        for (int i = threadIdx.x; i < A1 / 8; i += 256)                         // W1 fragments
            reinterpret_cast<fbbev_v4u*>(a1lds)[i] = reinterpret_cast<const fbbev_v4u*>(w1f)[i];
        fbbev_v4f acc2[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mt][r] = bias2[16 * mt + 4 * g + r];
        const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
        for (int st = 0; st < steps; ++st) {
            __syncthreads();        // tiles / W2 of frames 2 st, 2 st + 1 are ready; slots (2 st + 2) & 3, (2 st + 3) & 3 are free
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * st + h;
                if (t >= T1) break;                                             // block-uniform
                // no global load in this loop: W2_t and the bias of frame t were staged by the loader waves a step ahead
                const float* b1 = b1lds + (t & 3) * C;
                fbbev_v4f acc1[MT1];
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) acc1[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                const unsigned short* xrow = xt + (t & 3) * TILE + vl * XP;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bool ok = inb && 32 * s + 8 * g < C;
                    const fbbev_bf16x8 xv = fbbev_ld_bf16x8(xrow + (32 * s + 8 * g < C ? 32 * s + 8 * g : 0));   // clamped, zero selected at use
                    const fbbev_bf16x8 xo = ok ? xv : zero8;
#pragma unroll
                    for (int mt = 0; mt < MT1; ++mt)
                        acc1[mt] = fbbev_mfma_f32_16x16x32_bf16(fbbev_ld_bf16x8(a1lds + ((mt * KS + s) * 64 + lane) * 8), xo, acc1[mt]);
                }
                fbbev_wave_sync();                                              // the Y rows are wave-private
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) {
                    const fbbev_v4f bia = *reinterpret_cast<const fbbev_v4f*>(b1 + 16 * mt + 4 * g);          // LDS
                    const fbbev_v4f yv = {fmaxf(acc1[mt][0] + bia[0], 0.f), fmaxf(acc1[mt][1] + bia[1], 0.f),
                                          fmaxf(acc1[mt][2] + bia[2], 0.f), fmaxf(acc1[mt][3] + bia[3], 0.f)};
                    const fbbev_bf16x8 pk = fbbev_cvt_bf16x8(yv, yv);
                    unsigned long long four;
                    __builtin_memcpy(&four, &pk, 8);
                    *reinterpret_cast<unsigned long long*>(yrow + 16 * mt + 4 * g) = four;
                }
                fbbev_wave_sync();
                const unsigned short* a2t = a2buf + (t & 3) * A2S;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const fbbev_bf16x8 yo = fbbev_ld_bf16x8(yrow + 32 * s + 8 * g);
#pragma unroll
                    for (int mt = 0; mt < MT2; ++mt)
                        acc2[mt] = fbbev_mfma_f32_16x16x32_bf16(fbbev_ld_bf16x8(a2t + ((mt * KS + s) * 64 + lane) * 8), yo, acc2[mt]);
                }
            }
        }
        if (inb) {
            float* ob = out + (long long)b * Cout * N + n;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long long)(16 * mt + 4 * g + r) * N] = fmaxf(acc2[mt][r], 0.f);
        }
    } else {
        // ------------------------------------------------------------------ the warp (k_history_warp_vm's item body, two frames at a time)
        const int item = (int)threadIdx.x - 256;                              // 768 producer threads, 640 items
        const int vl = item / (C / 8), gq = item - vl * (C / 8);
        const int x = xc * 64 + vl;
        const bool act = item < ITEMS && x < X;
        const int xs = act ? x : xc * 64;                                     // clamped: unconditional loads, nothing stored
        unsigned int ob[8];
        float w[8];
        fbbev_warp_taps_vm(flow + b * 16, xs, y, z, X, Y, Z, C, (act ? gq : 0) * 8, 2, ob, w);
        const unsigned int lofs = (unsigned int)((act ? vl : 0) * XP + (act ? gq : 0) * 8);
        const unsigned int rofs = (unsigned int)((((long long)z * Y + y) * X + xs) * C + (act ? gq : 0) * 8) * 2u;
        const char* src = static_cast<const char*>(hist) + (size_t)b * hist_stride_b * 2;
        char* dst = static_cast<char*>(nxt) + (size_t)b * nxt_stride_b * 2;
        // frame f of the next ring (f >= 1) = frame f - 1 of the previous ring, re-sampled; frame 0 = the current frame, which
        // fbbev_history_frame_vm has already stored: only its operand pieces are copied into tile 0
        // (a build that issued the taps of pair p + 1 BEFORE the barrier ending pair p measured the same 7.1 ms and spilled: the
        // tap registers are then live across the barrier)
        auto produce_pair = [&](int f0) {                                      // frames f0 (even), f0 + 1
            fbbev_v4u raw[2][8];
            const int fa = f0 >= 1 ? f0 - 1 : 0, fb = f0 < T1 - 1 ? f0 : (T1 >= 2 ? T1 - 2 : 0);       // clamped source frames
            const char* sa = src + (size_t)fa * frame_bytes;
            const char* sb = src + (size_t)fb * frame_bytes;
            if (f0 == 0) {
                raw[0][0] = *reinterpret_cast<const fbbev_v4u*>(dst + rofs);  // frame 0: the current frame's stored piece
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) raw[0][q] = *reinterpret_cast<const fbbev_v4u*>(sa + ob[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) raw[1][q] = *reinterpret_cast<const fbbev_v4u*>(sb + ob[q]);
            fbbev_sched_fence();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f = f0 + h;
                if (f >= T1) break;                                            // block-uniform
                fbbev_v4u pk;
                if (f == 0) {
                    pk = raw[0][0];
                } else {
                    float acc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float a[8];
                        fbbev_widen_vec<ET>(raw[h][q], a);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], w[q], acc[e]);
                    }
                    pk = fbbev_narrow_vec<ET>(acc);
                }
                if (act) {
                    if (f > 0) {
                        fbbev_v4f pf;
                        __builtin_memcpy(&pf, &pk, 16);
                        fbbev_store4<1>(reinterpret_cast<float*>(dst + (size_t)f * frame_bytes + rofs), pf);
                    }
                    *reinterpret_cast<fbbev_v4u*>(xt + (f & 3) * TILE + lofs) = fbbev_hf_operand<ET>(pk);
                }
            }
        };
        // the two waves without items (threads 640..767 of the role) are the LOADERS: W2 fragments and conv-1 bias of a frame
        // pair, global -> registers -> LDS, a step ahead -- the consumers, one wave per SIMD, then never wait on a global load
        // (in the first build each frame cost them one full memory latency under the producers' traffic)
        const int ld = item - ITEMS;
        auto stage_pair = [&](int f0) {
            constexpr int R = (A2 / 8 + 127) / 128;                             // 16-byte pieces of one frame per loader thread
            for (int h = 0; h < 2; ++h) {
                const int f = f0 + h;
                if (f >= T1) break;
                fbbev_v4u wv[R];
                const fbbev_v4u* wn = reinterpret_cast<const fbbev_v4u*>(w2f + (long long)f * A2);
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int i = ld + 128 * q;
                    wv[q] = wn[i < A2 / 8 ? i : 0];
                }
                const float bv = bias1[((long long)b * T1 + f) * C + (ld < C ? ld : 0)];
                fbbev_v4u* wd = reinterpret_cast<fbbev_v4u*>(a2buf + (f & 3) * A2S);
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int i = ld + 128 * q;
                    if (i < A2 / 8) wd[i] = wv[q];
                }
                if (ld < C) b1lds[(f & 3) * C + ld] = bv;
            }
        };
        if (ld >= 0) {
            stage_pair(0);
            for (int st = 0; st < steps; ++st) {
                __syncthreads();
                if (st + 1 < steps) stage_pair(2 * (st + 1));
            }
        } else {
            produce_pair(0);
            for (int st = 0; st < steps; ++st) {
                __syncthreads();
                if (st + 1 < steps) produce_pair(2 * (st + 1));
            }
        }
    }
}
