// da_fused_kernels.h -- depth-aware spatial cross-attention, inference, ONE kernel from the BEV query rows to the slots
// (round 4; SURVEY 8b names it `fbbev_da_cross_attn_fused`).  Replaces k_rows_linear_x3 (sampling_offsets) + k_rows_linear_x3
// (attention_weights) + the softmax + k_da_cross_attn_fwd_pipe, i.e. spatial_cross_attention_depth.py:533-595 and :136-223.
//
// What was wrong with the three-kernel form at BASELINE configs[2] (B = 4, 200 x 200 queries, 4 levels; profiles/r03_*):
//   * the two projections wrote offsets (B,Q,8,4,8,2) + weights (B,Q,8,4,8) = 492 MB which the sampler read back: 0.26 ms of store
//     streams + the re-read;
//   * the sampler was bound by vector-L1 ACCESSES (456 M per launch, ~43 lines per load instruction): with token rows
//     [token][chunk][head][4 floats] a (token, head) piece is 16 bytes inside a 384-byte row, so two lanes share a line only when
//     they sample the SAME token.
// What this kernel does instead:
//   * HEAD-PLANE tokens: value_proj writes (B*Ncam, M, S, DH) -- the tokens of one head contiguous, DH floats each (40 bytes at
//     DH = 10, no padding).  The two x-corners of a bilinear sample are ONE 2*DH-float run (5 sixteen-byte loads instead of 6
//     scattered pieces), and lanes that sample neighbouring tokens of the same head share 128-byte lines;
//   * a WAVE owns ONE head of an 8 x 8 patch of BEV queries (a workgroup = the M heads of the patch): all 64 lanes read the
//     same head plane around the same image region -- tools/micro/plane_sampler.hip: 0.56 -> 0.34 ms per 41 M samples against
//     the row layout (profiles/r04_exp_plane_sampler.jsonl);
//   * the projections run INSIDE the workgroup on the bf16 MFMA with split operands (the arithmetic of k_rows_linear_x3:
//     v = hi + lo, three MFMAs per product, ~1e-5 relative): the 64 query rows (+ positional rows) are split once into LDS
//     fragments; each wave computes the logits of its head (L*P outputs -> softmax in LDS) and, level by level, the 2*P offsets
//     of its head, straight into a wave-private LDS tile -- nothing per-query ever goes to HBM, and the weight fragments
//     (2 x 123 KB in all, 6-12 KB per wave and step) stream from L2;
//   * the camera hit test / reference points / depth weights are per (query, camera), not per head: computed ONCE per
//     workgroup into LDS instead of once per head lane (8x fewer depth-plane samples).
// Loop order: level outer (its offsets live in 16 registers), camera inner (uniform loop over the cameras some lane of the
// wave hits), P samples per (level, camera) with two samples in flight per lane.  The reference sums per camera first
// (levels, points), then over cameras; here one accumulator takes (level, camera, point) order: equal up to fp32
// re-association (tests: <= 1e-4 against the oracle composite, observed ~1e-6).
// Padded corners: their WEIGHT is zeroed and the run is clamped into the row (w * 0 of the reference becomes 0 * v: the same
// +-0 contribution for finite tokens); an out-of-image or non-hit sample loads the level's first tokens with weight 0.
// Preconditions (launcher): M <= 8 waves, E = M*DH, DH in {8, 10}, P == 8, Za == 4, every level at least 2 tokens wide,
// Q = bev_h x bev_w, LDS budget (see fbbev_daf_lds_bytes).
#pragma once
#include "rt.h"
#include "da_kernels.h"
#include "rows_linear_kernels.h"

#define FBBEV_DAF_QC 13          // floats of a (camera, query) record: rx[4] ry[4] dw[4] hit
#define FBBEV_DAF_OS 20          // floats per query of a wave's offsets tile (16 used; 20 makes the 16-byte row reads conflict free)
#define FBBEV_DAF_P 8            // sampling points per level
#define FBBEV_DAF_ZA 4           // Z anchors per pillar

#define FBBEV_DAF_MAXL 4         // levels whose attention weights a lane keeps in registers (8 each)

// LDS carve-up (bytes): x fragments | (camera, query) records | per-wave transposition tiles (HW = heads = waves of a workgroup)
__host__ __device__ inline size_t fbbev_daf_xf_bytes(int E) { return (size_t)4 * ((E + 31) / 32) * 2 * 64 * 16; }
// stage_floats > 0 (round 5): a wave's transposition tile doubles as the staging buffer of a COARSE level's head plane (see the
// kernel): the per-wave region grows to max(tile, stage_floats) floats
__host__ __device__ inline int fbbev_daf_wave_region(int stage_floats) {
    const int t = 64 * FBBEV_DAF_OS;
    return (((stage_floats > t ? stage_floats : t) + 3) / 4) * 4;
}
__host__ __device__ inline size_t fbbev_daf_lds_bytes(int E, int HW, int Ncam, int stage_floats = 0) {
    return fbbev_daf_xf_bytes(E) + (size_t)Ncam * 64 * FBBEV_DAF_QC * 4 + (size_t)HW * fbbev_daf_wave_region(stage_floats) * 4;
}

// ET: element type of the head planes -- 0 fp32 (the default: the reference's precision), 1 bf16, 2 fp16 (round 5: the camera-token
// STORAGE option on head planes; every product and sum stays fp32).  A 16-bit run of two tokens is 2 DH halves = DH 32-bit words.
template <int DH, int ET = 0>
struct fbbev_daf_pending {
    static constexpr int NV = (2 * DH) / 4;
    fbbev_v4f a[ET == 0 ? NV : 1], b[ET == 0 ? NV : 1];   // fp32: the two row runs, tokens (x, x+1) of rows y0 and y1
    unsigned int ra[ET == 0 ? 1 : DH], rb[ET == 0 ? 1 : DH];   // 16-bit: the same runs as raw words (two channels each)
    float w00, w01, w10, w11, weight;
};

// run `k` float of slot `slot` (0 / 1) of a run
template <int DH>
__device__ __forceinline__ fbbev_v2f fbbev_daf_pair(const fbbev_v4f (&r)[(2 * DH) / 4], int slot, int c) {
    const int i = slot * DH + c;
    fbbev_v2f v;
    v[0] = r[i >> 2][i & 3]; v[1] = r[(i + 1) >> 2][(i + 1) & 3];
    return v;
}
template <int ET>
__device__ __forceinline__ float fbbev_daf_widen(unsigned int h) {          // 16 bits -> fp32, exact (1 bf16, 2 fp16)
    if constexpr (ET == 2) return fbbev_f16_bits_to_f32(h);
    else { const unsigned int u = h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
}
// channels (c, c + 1) of token `slot` of a raw 16-bit run, widened exactly
template <int DH, int ET>
__device__ __forceinline__ fbbev_v2f fbbev_daf_pair16(const unsigned int (&r)[DH], int slot, int c) {
    const unsigned int w = r[slot * (DH / 2) + (c >> 1)];
    fbbev_v2f v;
    v[0] = fbbev_daf_widen<ET>(w & 0xffffu); v[1] = fbbev_daf_widen<ET>(w >> 16);
    return v;
}

template <int DH, bool LDS = false, int ET = 0>
__device__ __forceinline__ void fbbev_daf_issue(const char* __restrict__ plane, int level_off /* elements */, float h_im, float w_im,
                                                int sh, int sw, float weight, bool enable, fbbev_daf_pending<DH, ET>& p) {
    const bool live = enable && h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw;
    const float h = live ? h_im : 0.f, w = live ? w_im : 0.f;
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const float lh = h - (float)h_low, lw = w - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    // x: the run covers tokens (xb, xb + 1), clamped into the row; at the left edge the only valid corner (x = 0, the HIGH
    // corner) sits in slot 0, at the right edge (x = W - 1, the LOW corner) in slot 1
    const bool left = w_low < 0, right = w_low >= sw - 1;
    const int xb = left ? 0 : (right ? sw - 2 : w_low);
    const float sx0 = left ? lw : (right ? 0.f : hw), sx1 = left ? 0.f : (right ? hw : lw);
    // y: an invalid row reads the valid one again (the same lines) with weight 0
    const bool top = h_low < 0, bottom = h_low >= sh - 1;
    const int y0 = top ? 0 : h_low, y1 = bottom ? h_low : h_low + 1;
    const float sy0 = top ? 0.f : hh, sy1 = bottom ? 0.f : lh;
    p.w00 = sy0 * sx0; p.w01 = sy0 * sx1; p.w10 = sy1 * sx0; p.w11 = sy1 * sx1;
    p.weight = live ? weight : 0.f;
    constexpr unsigned ESZ = ET == 0 ? 4u : 2u;
    // byte offsets of the two runs inside the head plane: token index < S < 2^24 (the launcher checks), 24-bit multiplies (full rate)
    const unsigned o0 = fbbev_mad_u24_vks<DH * ESZ>(fbbev_mad_u24_vsv((unsigned)y0, (unsigned)sw, (unsigned)xb), (unsigned)level_off * ESZ);
    const unsigned o1 = fbbev_mad_u24_vks<DH * ESZ>(fbbev_mad_u24_vsv((unsigned)y1, (unsigned)sw, (unsigned)xb), (unsigned)level_off * ESZ);
    if constexpr (ET == 0) {
        constexpr int NV = (2 * DH) / 4;
#pragma unroll
        for (int k = 0; k < NV; ++k) {          // DH = 10: 8-byte aligned 16-byte loads (global memory takes dword-aligned b128)
            if constexpr (LDS) {                // the level's plane staged in LDS: no vector-L1 access at all
                p.a[k] = fbbev_lds_ld_v4f_a8(reinterpret_cast<const float*>(plane + o0 + 16 * k));
                p.b[k] = fbbev_lds_ld_v4f_a8(reinterpret_cast<const float*>(plane + o1 + 16 * k));
            } else {
                fbbev_v4f t0, t1;
                __builtin_memcpy(&t0, plane + o0 + 16 * k, 16);
                __builtin_memcpy(&t1, plane + o1 + 16 * k, 16);
                p.a[k] = t0; p.b[k] = t1;
            }
        }
    } else {                                    // 16-bit tokens: DH words per run, 4-byte aligned (a token is 2 DH bytes)
#pragma unroll
        for (int k = 0; k < (LDS ? DH : 0); k += 2) {
            if constexpr (LDS) {
                p.ra[k] = (unsigned int)fbbev_lds_ld_i32(reinterpret_cast<const int*>(plane + o0 + 4 * k));
                p.ra[k + 1] = (unsigned int)fbbev_lds_ld_i32(reinterpret_cast<const int*>(plane + o0 + 4 * k + 4));
                p.rb[k] = (unsigned int)fbbev_lds_ld_i32(reinterpret_cast<const int*>(plane + o1 + 4 * k));
                p.rb[k + 1] = (unsigned int)fbbev_lds_ld_i32(reinterpret_cast<const int*>(plane + o1 + 4 * k + 4));
            }
        }
        if constexpr (!LDS) {                   // global: DH = 10 -> b128, b128, b64 per row at a dword-aligned address (3 loads, fp32: 5)
            constexpr int N4 = DH / 4 * 4;
#pragma unroll
            for (int k = 0; k < N4; k += 4) {
                unsigned int t0[4], t1[4];
                __builtin_memcpy(t0, plane + o0 + 4 * k, 16);
                __builtin_memcpy(t1, plane + o1 + 4 * k, 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) { p.ra[k + i] = t0[i]; p.rb[k + i] = t1[i]; }
            }
            if constexpr (N4 < DH) {
                unsigned int t0[2], t1[2];
                __builtin_memcpy(t0, plane + o0 + 4 * N4, 8);
                __builtin_memcpy(t1, plane + o1 + 4 * N4, 8);
                p.ra[N4] = t0[0]; p.ra[N4 + 1] = t0[1]; p.rb[N4] = t1[0]; p.rb[N4 + 1] = t1[1];
            }
        }
    }
}

template <int DH, int ET = 0>
__device__ __forceinline__ void fbbev_daf_consume(const fbbev_daf_pending<DH, ET>& p, fbbev_v2f (&col)[DH / 2]) {
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        fbbev_v2f v00, v01, v10, v11;
        if constexpr (ET == 0) {
            v00 = fbbev_daf_pair<DH>(p.a, 0, c); v01 = fbbev_daf_pair<DH>(p.a, 1, c);
            v10 = fbbev_daf_pair<DH>(p.b, 0, c); v11 = fbbev_daf_pair<DH>(p.b, 1, c);
        } else {
            v00 = fbbev_daf_pair16<DH, ET>(p.ra, 0, c); v01 = fbbev_daf_pair16<DH, ET>(p.ra, 1, c);
            v10 = fbbev_daf_pair16<DH, ET>(p.rb, 0, c); v11 = fbbev_daf_pair16<DH, ET>(p.rb, 1, c);
        }
        col[c / 2] += (p.w00 * v00 + p.w01 * v01 + p.w10 * v10 + p.w11 * v11) * p.weight;
    }
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) fbbev_pin(col[c]);
}

// one 16-output tile of a split-operand projection of the workgroup's 64 query rows: acc[rt] (row tile rt = queries 16 rt ..
// 16 rt + 15) = W[16 T .. 16 T + 15][:] . x^T, lane (g, j) holds outputs 16 T + 4 g + r (r = 0..3) of query 16 rt + j
// (two steps: the tile's six weight fragments are REQUESTED first -- fbbev_daf_wload, 24 registers that are free while no sample is
// in flight -- so that the caller can put its other loads behind them and the MFMAs wait for one round trip only)
template <int KS>
struct fbbev_daf_wtile { fbbev_bf16x8 h[KS], l[KS]; };
template <int KS>
__device__ __forceinline__ void fbbev_daf_wload(const unsigned short* __restrict__ wf, int T, int lane, fbbev_daf_wtile<KS>& w) {
    const unsigned short* wt = wf + (long long)T * FBBEV_RL_TILE_ELEMS;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        w.h[s] = fbbev_ld_bf16x8(wt + (s * 64 + lane) * 8);
        w.l[s] = fbbev_ld_bf16x8(wt + FBBEV_RL_TILE_ELEMS / 2 + (s * 64 + lane) * 8);
    }
}
template <int KS>
__device__ __forceinline__ void fbbev_daf_project_w(const fbbev_daf_wtile<KS>& w, const unsigned short* __restrict__ xf, int lane,
                                                    fbbev_v4f (&acc)[4]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const fbbev_bf16x8 xh = fbbev_ld_bf16x8(xf + (((rt * KS + s) * 2 + 0) * 64 + lane) * 8);
            const fbbev_bf16x8 xl = fbbev_ld_bf16x8(xf + (((rt * KS + s) * 2 + 1) * 64 + lane) * 8);
            acc[rt] = fbbev_mfma_f32_16x16x32_bf16(w.l[s], xh, acc[rt]);
            acc[rt] = fbbev_mfma_f32_16x16x32_bf16(w.h[s], xl, acc[rt]);
            acc[rt] = fbbev_mfma_f32_16x16x32_bf16(w.h[s], xh, acc[rt]);
        }
    }
}
template <int KS>
__device__ __forceinline__ void fbbev_daf_project(const unsigned short* __restrict__ wf, int T, const unsigned short* __restrict__ xf,
                                                  int lane, fbbev_v4f (&acc)[4]) {
    fbbev_daf_wtile<KS> w;
    fbbev_daf_wload<KS>(wf, T, lane, w);
    fbbev_daf_project_w<KS>(w, xf, lane, acc);
}

// One bilinear sample of a (H, W >= 2) plane at normalised (x, y) as four CLAMPED corner offsets + weights: a padded corner keeps a
// valid address and gets weight 0, a sample outside the image gets four zero weights (fbbev_plane_sample's value: the reference's
// w * 0 for a padded corner becomes 0 * v), so the caller can issue all loads unconditionally.
__device__ __forceinline__ void fbbev_daf_plane_corners(float x, float y, int H, int W, int (&off)[4], float (&wgt)[4]) {
    const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
    const bool live = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float h = live ? h_im : 0.f, w = live ? w_im : 0.f;
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const float lh = h - (float)h_low, lw = w - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const bool left = w_low < 0, right = w_low >= W - 1, top = h_low < 0, bottom = h_low >= H - 1;
    const int xa = left ? 0 : w_low, xb = right ? w_low : w_low + 1;       // (low, high) x corners, clamped
    const int ya = top ? 0 : h_low, yb = bottom ? h_low : h_low + 1;
    const float s = live ? 1.f : 0.f;
    off[0] = ya * W + xa; off[1] = ya * W + xb; off[2] = yb * W + xa; off[3] = yb * W + xb;
    wgt[0] = (top || left) ? 0.f : s * hh * hw; wgt[1] = (top || right) ? 0.f : s * hh * lw;
    wgt[2] = (bottom || left) ? 0.f : s * lh * hw; wgt[3] = (bottom || right) ? 0.f : s * lh * lw;
}

// ---------------------------------------------------------------- output projection + residual + LayerNorm in the workgroup
// The attention output of a patch never leaves the CU: every wave drops its head's DH channels of the 64 queries into the
// workgroup's x-fragment buffer (split bf16, the B-operand layout of fbbev_daf_project -- the query fragments are no longer
// needed), then wave rt (rt = 0..3) takes the 16 queries of row tile rt through `output_proj` (O = 16 MT outputs, the
// three-MFMA split-operand arithmetic of k_rows_linear_x3), adds bias + residual and applies the layer's LayerNorm exactly as
// k_rows_linear_x3<., true>'s epilogue does (two-pass statistics; a row's outputs live in the 4 lanes g = 0..3 of lane j).
// Replaces a k_rows_linear_x3_ln launch and the write + read of the (B, Q, E) attention output between the two kernels.
struct fbbev_daf_outproj {
    const unsigned short* w_frag;      // output_proj.weight as split bf16 fragments (k_rows_linear_x3_fragments); null = no epilogue
    const float* bias;                 // (O)
    const float* res;                  // residual rows (B*Q, ld_res), may be null
    long long ld_res;
    const float* ln_w;                 // LayerNorm weight / bias (O)
    const float* ln_b;
    float eps;
};

template <int DH, int KS>
__device__ __forceinline__ void fbbev_daf_put_channels(unsigned short* __restrict__ xf, int m, int lane, const fbbev_v2f (&acc)[DH / 2]) {
    static_assert(DH <= 16, "two groups of eight");
    const int rt = lane >> 4, j = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // eight channels at a time (registers: the sampler's budget is the kernel's)
        if (8 * h >= DH) break;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 8 * h + c < DH ? acc[(8 * h + c) >> 1][c & 1] : 0.f;
        fbbev_bf16x8 h8, l8;
        fbbev_split_bf16x8(fbbev_v4f{v[0], v[1], v[2], v[3]}, fbbev_v4f{v[4], v[5], v[6], v[7]}, h8, l8);
        unsigned short hs[8], ls[8];
        __builtin_memcpy(hs, &h8, 16);
        __builtin_memcpy(ls, &l8, 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (8 * h + c >= DH) break;
            const int ch = m * DH + 8 * h + c, s = ch >> 5, g = (ch & 31) >> 3, e = ch & 7;            // uniform over the wave
            unsigned short* dst = xf + (((rt * KS + s) * 2) * 64 + g * 16 + j) * 8 + e;
            dst[0] = hs[c];
            dst[64 * 8] = ls[c];
        }
    }
}

// rows 16 rt .. 16 rt + 15 of the patch (row of lane (g, j): `row`, `live`): out[row][:O] = LN(W x + b + res[row])
template <int KS, int MT>
__device__ __forceinline__ void fbbev_daf_outproj_ln(const fbbev_daf_outproj& op, const unsigned short* __restrict__ xf, int rt, int lane,
                                                     long long row, bool live, float* __restrict__ out, long long ldo) {
    constexpr int O = 16 * MT;
    const int g = lane >> 4;
    // bias and residual pieces requested ahead of the MFMAs, unconditionally (`row` is valid for a dead lane too): behind their `if`
    // every piece was its own round trip at the tail of the workgroup (round 5)
    fbbev_v4f v[MT], pb[MT], pr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pb[mt] = *reinterpret_cast<const fbbev_v4f*>(op.bias + 16 * mt + 4 * g);
    if (op.res) {                                                       // uniform
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pr[mt] = fbbev_gld_v4f(op.res + row * op.ld_res + 16 * mt + 4 * g);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) v[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const fbbev_bf16x8 xh = fbbev_ld_bf16x8(xf + (((rt * KS + s) * 2 + 0) * 64 + lane) * 8);
        const fbbev_bf16x8 xl = fbbev_ld_bf16x8(xf + (((rt * KS + s) * 2 + 1) * 64 + lane) * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned short* wt = op.w_frag + (long long)mt * FBBEV_RL_TILE_ELEMS;
            const fbbev_bf16x8 ah = fbbev_ld_bf16x8(wt + (s * 64 + lane) * 8);
            const fbbev_bf16x8 al = fbbev_ld_bf16x8(wt + FBBEV_RL_TILE_ELEMS / 2 + (s * 64 + lane) * 8);
            v[mt] = fbbev_mfma_f32_16x16x32_bf16(al, xh, v[mt]);
            v[mt] = fbbev_mfma_f32_16x16x32_bf16(ah, xl, v[mt]);
            v[mt] = fbbev_mfma_f32_16x16x32_bf16(ah, xh, v[mt]);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        v[mt] = v[mt] + pb[mt];
        if (op.res && live) v[mt] = v[mt] + pr[mt];
        sum += (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
    }
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)O;
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        v[mt] = v[mt] - fbbev_v4f{mean, mean, mean, mean};
        q += (v[mt][0] * v[mt][0] + v[mt][1] * v[mt][1]) + (v[mt][2] * v[mt][2] + v[mt][3] * v[mt][3]);
    }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float inv = 1.0f / sqrtf(q / (float)O + op.eps);
    if (!live) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {                                   // (pb / pr are free: the LayerNorm pieces, requested together)
        const int o = 16 * mt + 4 * g;
        pb[mt] = *reinterpret_cast<const fbbev_v4f*>(op.ln_w + o);
        pr[mt] = *reinterpret_cast<const fbbev_v4f*>(op.ln_b + o);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int o = 16 * mt + 4 * g;
        const fbbev_v4f w4 = pb[mt], b4 = pr[mt];
        fbbev_v4f y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = v[mt][e] * inv * w4[e] + b4[e];
        fbbev_st(reinterpret_cast<fbbev_v4f*>(out + row * ldo + o), y);
    }
}

// the 64 query rows (+ positional rows) of an 8 x 8 patch as split bf16 B fragments in LDS: [4 row tiles][KS][hi|lo][64][8].  Every
// piece of the workgroup is REQUESTED before the first is split and stored (round 5: as a `for (i = tid; ...; i += NT)` loop the
// compiler waited for each iteration's loads before the next iteration's were issued -- three round trips to the query rows in a row
// at 256 threads)
template <int E, int KS, int NT>
__device__ __forceinline__ void fbbev_daf_query_fragments(unsigned short* __restrict__ xf, const float* __restrict__ query, long long ldq,
                                                          const float* __restrict__ addend, long long ld_add, long long add_period,
                                                          int b, int Q, int bev_w, int bev_h, int x0, int y0) {
    constexpr int N = 4 * KS * 64, NI = (N + NT - 1) / NT;
    fbbev_v4f q4[NI][2], a4[NI][2];
    bool ok[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = (int)threadIdx.x + NT * k;
        const int rt = i / (KS * 64), s = (i >> 6) % KS, ln = i & 63;
        const int g = ln >> 4, j = ln & 15, ql = 16 * rt + j, c = 32 * s + 8 * g;
        const int qy = y0 + (ql >> 3), qx = x0 + (ql & 7);
        ok[k] = i < N && c < E && qy < bev_h && qx < bev_w;
        const long long row = ok[k] ? (long long)b * Q + (long long)qy * bev_w + qx : (long long)b * Q;      // (clamped: unconditional loads)
        const float* src = query + row * ldq + (ok[k] ? c : 0);
        q4[k][0] = *reinterpret_cast<const fbbev_v4f*>(src); q4[k][1] = *reinterpret_cast<const fbbev_v4f*>(src + 4);
        if (addend) {                                                                       // uniform
            const float* a = addend + (row % add_period) * ld_add + (ok[k] ? c : 0);
            a4[k][0] = *reinterpret_cast<const fbbev_v4f*>(a); a4[k][1] = *reinterpret_cast<const fbbev_v4f*>(a + 4);
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = (int)threadIdx.x + NT * k;
        if (i >= N) break;
        const int rt = i / (KS * 64), s = (i >> 6) % KS, ln = i & 63;
        fbbev_v4f lo4 = q4[k][0], hi4 = q4[k][1];
        if (addend) { lo4 = lo4 + a4[k][0]; hi4 = hi4 + a4[k][1]; }
        if (!ok[k]) { lo4 = fbbev_v4f{0.f, 0.f, 0.f, 0.f}; hi4 = fbbev_v4f{0.f, 0.f, 0.f, 0.f}; }
        fbbev_bf16x8 h8, l8;
        fbbev_split_bf16x8(lo4, hi4, h8, l8);
        __builtin_memcpy(xf + (((rt * KS + s) * 2 + 0) * 64 + ln) * 8, &h8, 16);
        __builtin_memcpy(xf + (((rt * KS + s) * 2 + 1) * 64 + ln) * 8, &l8, 16);
    }
}

// planes (B*Ncam, M, S, DH); pred_depth (B*Ncam, DC, H0, W0); ref_cam (Ncam,B,Q,Za,2); mask (Ncam,B,Q,Za) u8; qdepth (Ncam,B,Q,Za);
// query (B*Q rows, ldq floats apart, E used) [+ addend rows: row (b*Q + q) % add_period]; so_frag / aw_frag: split bf16 fragments of
// sampling_offsets.weight (M*L*P*2, E) / attention_weights.weight (M*L*P, E) in the order of k_rows_linear_x3_fragments, rows in
// the MODULE's order ((m, l, p, xy) / (m, l, p)); so_bias / aw_bias fp32; slots (B, Q, M*DH).  blockDim = 64 * M.
template <int DH, int MH, int NP, int HW, bool OP = false, int ET = 0, bool DG = false>
__global__ void __launch_bounds__(64 * HW, 2)     // two waves per SIMD (HW = 4: two workgroups per CU; HW = 8: one): at most 256 registers
k_da_cross_attn_fused(const float* __restrict__ planes /* ET != 0: 16-bit elements behind the pointer */, const int64_t* __restrict__ spatial_shapes,
                      const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                      const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                      const float* __restrict__ qdepth, const float* __restrict__ query, long long ldq,
                      const float* __restrict__ addend, long long ld_add, long long add_period,
                      const unsigned short* __restrict__ so_frag, const float* __restrict__ so_bias,
                      const unsigned short* __restrict__ aw_frag, const float* __restrict__ aw_bias,
                      int B, int Ncam, int S, int L, int Q, int bev_w, int DC, float d0, float dstep, float* __restrict__ slots,
                      fbbev_daf_outproj op, int stage_floats) {
    constexpr int E = MH * DH, KS = (E + 31) / 32, P = FBBEV_DAF_P, ZA = FBBEV_DAF_ZA, NT = 64 * HW, PARTS = MH / HW;
    const bool pre_copy = (stage_floats & 0x40000000) == 0;     // A/B bit of the launcher: the first staged copy ahead of the projection
    // timing diagnostics of the launcher (FBBEV_DA_FUSED_DIAG, results are WRONG): 1 = no samples at all (phase A + projections +
    // softmax remain), 2 = every sample reads token 0 of its level (the instruction stream without the gather's divergence)
    // 4 = no (camera, query) records (nothing hits), 8 = no projections (MFMA tiles skipped)
    // (DG: the diagnostic instantiation -- the bits cost registers; the product kernels compile them out)
    // exit points (diag >> 4): 1 = return behind phase A, 2 = behind the softmax, 3 = no final store
    const int diag = DG ? ((stage_floats >> 24) & 63) : 0;
    stage_floats &= 0x00ffffff;
    static_assert(!OP || (HW == MH && E % 16 == 0), "the output_proj + LayerNorm tail needs all heads of a query in one workgroup");
    static_assert((2 * DH) % 4 == 0 && DH % 2 == 0 && E % 8 == 0, "runs of whole 16-byte pieces, channel pairs");
    static_assert(NP >= 2 && NP <= FBBEV_DAF_P, "samples in flight per lane");
    static_assert(MH % HW == 0 && P == 8, "a workgroup takes HW of the MH heads; a level's 8 logits are half an MFMA tile");
    unsigned char* lds = reinterpret_cast<unsigned char*>(fbbev_dyn_lds_f32());
    unsigned short* xf = reinterpret_cast<unsigned short*>(lds);                            // [4][KS][hi|lo][64][8] bf16
    float* qc = reinterpret_cast<float*>(lds + fbbev_daf_xf_bytes(E));                      // [Ncam][64][QC]
    float* off_all = qc + (size_t)Ncam * 64 * FBBEV_DAF_QC;                                 // [HW][64][OS]
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int bev_h = Q / bev_w;
    const int pxn = (bev_w + 7) / 8, pyn = (bev_h + 7) / 8;
    const long long n_wg = (long long)B * pxn * pyn * PARTS, per_xcd = (n_wg + 7) / 8;
    const long long wgid_ = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // XCD-contiguous patch order
    if (wgid_ >= n_wg) return;                                                              // uniform
    // PARTS workgroups share a patch (HW heads each: their own copy of phase A, so that one's prologue runs under the other's
    // samples on the same CU); consecutive ids = the same XCD
    const long long wgid = wgid_ / PARTS;
    const int part = (int)(wgid_ - wgid * PARTS);
    const int b = (int)(wgid / ((long long)pxn * pyn)), pi = (int)(wgid - (long long)b * pxn * pyn);
    const int py = pi / pxn, px = pi - py * pxn;
    const int x0 = px * 8, y0 = py * 8;
    // ---------------- phase A (whole workgroup): the patch's query rows as split MFMA fragments, its (camera, query) records
    // hit flags first (one round trip, under the query rows' loads below): a record of a camera the query does not see -- three of
    // four at six surround cameras -- is never computed (round 5; before, every record paid its 16 depth loads and ~300 VALU)
    for (int i0 = threadIdx.x; i0 < ((diag & 4) ? 0 : Ncam * 64); i0 += 2 * NT) {           // (two records' masks per round trip)
        unsigned int mask4[2];
        bool inb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + NT * u < Ncam * 64 ? i0 + NT * u : i0;
            const int cam = i >> 6, ql = i & 63;
            const int qy = y0 + (ql >> 3), qx = x0 + (ql & 7);
            inb[u] = qy < bev_h && qx < bev_w;
            const long long base = (((long long)cam * B + b) * Q + (inb[u] ? (long long)qy * bev_w + qx : 0)) * ZA;
            __builtin_memcpy(&mask4[u], mask + base, 4);                                // ZA = 4 mask bytes
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (i0 + NT * u < Ncam * 64) qc[(size_t)(i0 + NT * u) * FBBEV_DAF_QC + 3 * ZA] = (inb[u] && mask4[u] != 0u) ? 1.f : 0.f;
    }
    fbbev_daf_query_fragments<E, KS, NT>(xf, query, ldq, addend, ld_add, add_period, b, Q, bev_w, bev_h, x0, y0);
    for (int i = threadIdx.x; i < ((diag & 4) ? 0 : Ncam * 64); i += NT) {
        // every load of a hit record is issued before any is used; a non-hit record keeps only its flag (phase B reads the other
        // fields of such a record only into selects that discard them)
        const int cam = i >> 6, ql = i & 63;
        const int qy = y0 + (ql >> 3), qx = x0 + (ql & 7);
        float* rec = qc + (size_t)i * FBBEV_DAF_QC;
        if (fbbev_lds_ld_f32(rec + 3 * ZA) == 0.f) continue;                            // (this thread's own flag: no barrier needed)
        const long long base = (((long long)cam * B + b) * Q + (long long)qy * bev_w + qx) * ZA;
        const fbbev_v4f r01 = *reinterpret_cast<const fbbev_v4f*>(ref_cam + base * 2);  // (x0, y0, x1, y1)
        const fbbev_v4f r23 = *reinterpret_cast<const fbbev_v4f*>(ref_cam + base * 2 + 4);
        const fbbev_v4f qd = *reinterpret_cast<const fbbev_v4f*>(qdepth + base);
        const long long bn = (long long)b * Ncam + cam;
        float rx[ZA], ry[ZA], wgt[ZA][4], val[ZA][4];
#pragma unroll
        for (int z = 0; z < ZA; ++z) {
            rx[z] = z < 2 ? r01[2 * z] : r23[2 * z - 4]; ry[z] = z < 2 ? r01[2 * z + 1] : r23[2 * z - 3];
            float fb = floorf(__fdiv_rn(__fsub_rn(qd[z], d0), dstep));
            fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
            const float* plane = pred_depth + (bn * DC + (int)fb) * (long long)(H0 * W0);
            int off[4];
            fbbev_daf_plane_corners(rx[z], ry[z], H0, W0, off, wgt[z]);
#pragma unroll
            for (int k = 0; k < 4; ++k) val[z][k] = plane[off[k]];
        }
#pragma unroll
        for (int z = 0; z < ZA; ++z) {
            rec[z] = rx[z]; rec[ZA + z] = ry[z];
            rec[2 * ZA + z] = wgt[z][0] * val[z][0] + wgt[z][1] * val[z][1] + wgt[z][2] * val[z][2] + wgt[z][3] * val[z][3];
        }
    }
    __syncthreads();
    if ((diag >> 4) == 1) return;
    // ---------------- phase B (per wave = head m, no workgroup barrier below)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = part * HW + wv;
    const int g = lane >> 4, j = lane & 15;
    float* off_w = off_all + (size_t)wv * fbbev_daf_wave_region(stage_floats);
    fbbev_v4f pacc[4];
    if (diag & 8)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) pacc[rt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    // logits of head m: the 8 logits of (head m, level l) are rows (m*L + l)*8 .. +7 of attention_weights = HALF of one
    // 16-output tile; the tile goes through the wave's transposition tile (the path of the offsets below) and the lane keeps
    // its 8 values in registers: lg[l][p], levels taken last to first while the rows shift up, so lg[0] is level 0
    float lg[FBBEV_DAF_MAXL][P];
#pragma unroll
    for (int k = 0; k < FBBEV_DAF_MAXL; ++k)
#pragma unroll
        for (int p = 0; p < P; ++p) lg[k][p] = 0.f;
    {
        int t_have = -1;
        // (the weight fragments of the NEXT logits tile are requested before the MFMAs of the current one)
        const int T_lo = (m * L) >> 1;
        fbbev_daf_wtile<KS> wcur, wnxt;
        fbbev_daf_wload<KS>(aw_frag, (m * L + L - 1) >> 1, lane, wcur);
        wnxt = wcur;
        for (int l = L - 1; l >= 0; --l) {
            const int G = m * L + l, T = G >> 1, h8 = (G & 1) * 8;
            if (T != t_have) {                                                              // uniform
                if (T - 1 >= T_lo) fbbev_daf_wload<KS>(aw_frag, T - 1, lane, wnxt);
                if (!(diag & 8)) fbbev_daf_project_w<KS>(wcur, xf, lane, pacc);
                wcur = wnxt;
                t_have = T;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) __builtin_memcpy(off_w + (16 * rt + j) * FBBEV_DAF_OS + 4 * g, &pacc[rt], 16);
                fbbev_wave_sync();
            }
            fbbev_v4f c0, c1;
            __builtin_memcpy(&c0, off_w + lane * FBBEV_DAF_OS + h8, 16);
            __builtin_memcpy(&c1, off_w + lane * FBBEV_DAF_OS + h8 + 4, 16);
            if (l == 0 || ((m * L + l - 1) >> 1) != T) fbbev_wave_sync();                   // the next level overwrites the tile
#pragma unroll
            for (int k = FBBEV_DAF_MAXL - 1; k > 0; --k)
#pragma unroll
                for (int p = 0; p < P; ++p) lg[k][p] = lg[k - 1][p];
#pragma unroll
            for (int p = 0; p < P; ++p) lg[0][p] = (p < 4 ? c0[p] : c1[p - 4]) + aw_bias[G * P + p];
        }
    }
    {   // softmax over the unit's L*P logits (spatial_cross_attention_depth.py:546-551): exp(x - max) / sum, as ATen's kernel
        float mx = lg[0][0];
#pragma unroll
        for (int k = 0; k < FBBEV_DAF_MAXL; ++k)
            if (k < L)
#pragma unroll
                for (int p = 0; p < P; ++p) mx = fmaxf(mx, lg[k][p]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < FBBEV_DAF_MAXL; ++k)
            if (k < L)
#pragma unroll
                for (int p = 0; p < P; ++p) { lg[k][p] = __expf(lg[k][p] - mx); sum += lg[k][p]; }
        const float inv_sum = 1.f / sum;
#pragma unroll
        for (int k = 0; k < FBBEV_DAF_MAXL; ++k)
#pragma unroll
            for (int p = 0; p < P; ++p) lg[k][p] *= inv_sum;
    }
    if ((diag >> 4) == 2) { if (lg[0][0] + lg[1][1] + lg[2][2] + lg[3][3] == 123.f) slots[0] = 0.f; return; }
    const int qy = y0 + (lane >> 3), qx = x0 + (lane & 7);
    const bool valid = qy < bev_h && qx < bev_w;
    const long long bq = (long long)b * Q + (long long)qy * bev_w + qx;
    const float* my_qc = qc + (size_t)lane * FBBEV_DAF_QC;
    // the lane's hit flags as a bit mask and the cameras ANY lane of the wave hits (uniform), read once: the level loop below walks
    // the set bits instead of reading a flag + ballot per (level, camera)  (Ncam <= 32: the launcher checks)
    unsigned hitmask = 0u;
    for (int cam = 0; cam < Ncam; ++cam)
        hitmask |= (fbbev_lds_ld_f32(my_qc + (size_t)cam * 64 * FBBEV_DAF_QC + 3 * ZA) != 0.f) ? (1u << cam) : 0u;
    const int count = __builtin_popcount(hitmask);
    unsigned wave_cams_ = 0u;
    for (int cam = 0; cam < Ncam; ++cam)
        if (__ballot((hitmask >> cam) & 1u) != 0ull) wave_cams_ |= 1u << cam;
    const unsigned wave_cams = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_cams_);
    fbbev_v2f acc[DH / 2];
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) { acc[c][0] = 0.f; acc[c][1] = 0.f; }
    const char* pb = reinterpret_cast<const char*>(planes);
    for (int l = 0; l < L; ++l) {
        // a staged level (see below): the copy of its plane for the FIRST hit camera is requested here, into registers that are
        // free during the projection (the sample slots are dead), and lands under the MFMAs instead of in front of the samples
        constexpr int ESZ = ET == 0 ? 4 : 2;
        const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
        const int lvl_off = (int)level_start[l] * DH;                                        // elements
        const int lvl_n = sh * sw * DH * ESZ / 4;                                            // 4-byte words of the level's plane (DH even)
        const bool staged = lvl_n <= stage_floats;                                           // uniform
        // the 2*P offsets of (head m, level l): rows ((m*L + l)*P + p)*2 + xy of sampling_offsets = ONE 16-output tile.  Its weight
        // fragments and bias are requested before anything else of the level (in-order vmcnt: the MFMAs wait for these only)
        const int T = m * L + l;
        fbbev_daf_wtile<KS> wso;
        if (!(diag & 8)) fbbev_daf_wload<KS>(so_frag, T, lane, wso);
        const fbbev_v4f bias4 = *reinterpret_cast<const fbbev_v4f*>(so_bias + 16 * T + 4 * g);
        constexpr int PRE_N = ET == 0 ? 7 : 4;                                               // 16-byte pieces per lane held ahead (7 x 256 words >= 1 760)
        fbbev_v4f pre[PRE_N];
        int cam0 = -1;
        bool pre_ok = false;
        if (!OP && staged && pre_copy) {       // (the tail instantiation has no registers to spare for the held pieces)
            if (wave_cams != 0u) cam0 = __builtin_ctz(wave_cams);
            if (cam0 >= 0) {
                const float* src = reinterpret_cast<const float*>(pb + ((((long long)b * Ncam + cam0) * MH + m) * (long long)S + level_start[l]) * DH * ESZ);
                pre_ok = (lvl_n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && lvl_n <= PRE_N * 256;   // uniform
                if (pre_ok) {
#pragma unroll
                    for (int k = 0; k < PRE_N; ++k) {
                        const int i = lane * 4 + 256 * k;
                        pre[k] = *reinterpret_cast<const fbbev_v4f*>(src + (i < lvl_n ? i : 0));
                    }
                }
            }
        }
        if (!(diag & 8)) fbbev_daf_project_w<KS>(wso, xf, lane, pacc);
        {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const fbbev_v4f v = pacc[rt] + bias4;
                __builtin_memcpy(off_w + (16 * rt + j) * FBBEV_DAF_OS + 4 * g, &v, 16);
            }
        }
        fbbev_wave_sync();
        fbbev_v4f o4[4];                         // this lane's offsets: o4[p / 2] = (x_p, y_p, x_{p+1}, y_{p+1})
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_memcpy(&o4[k], off_w + lane * FBBEV_DAF_OS + 4 * k, 16);
        fbbev_wave_sync();                       // every lane holds its offsets before the tile is overwritten
        const float fsh = (float)sh, fsw = (float)sw;
        // round 5: a level whose head plane fits the wave's staging region (the transposition tile, free between two projections,
        // + the launcher's extra bytes: 8 x 22 and 4 x 11 tokens at BASELINE configs[2]) is copied into LDS once per hit camera and
        // sampled from there: ds_read instead of ~22 vector-L1 line accesses per load instruction -- the counter that bounds
        // this kernel (TCP_TOTAL_CACHE_ACCESSES: 195 M per launch, 0.71 per CU-cycle; profiles/r04_pmc_fb_BL3_B4_final.json)
        for (unsigned rem = (diag & 1) ? 0u : wave_cams; rem != 0u; rem &= rem - 1u) {       // uniform: the cameras somebody in the wave hits
            const int cam = __builtin_ctz(rem);
            const float* rec = my_qc + (size_t)cam * 64 * FBBEV_DAF_QC;
            const bool hit = ((hitmask >> cam) & 1u) != 0u;
            float rx[ZA], ry[ZA], dw[ZA];
#pragma unroll
            for (int z = 0; z < ZA; ++z) {
                rx[z] = fbbev_lds_ld_f32(rec + z); ry[z] = fbbev_lds_ld_f32(rec + ZA + z); dw[z] = fbbev_lds_ld_f32(rec + 2 * ZA + z);
            }
            const char* plane = pb + (((long long)b * Ncam + cam) * MH + m) * (long long)S * DH * ESZ;   // wave-uniform base
            // NP samples in flight per lane: sample p + NP - 1 is issued before sample p is blended (register slots addressed
            // at compile time: the P samples are unrolled)
            fbbev_daf_pending<DH, ET> pend[NP];
            if (staged) {
                const float* src = reinterpret_cast<const float*>(plane + (long long)lvl_off * ESZ);
                if (cam == cam0 && pre_ok) {                                                 // requested before the projection
#pragma unroll
                    for (int k = 0; k < PRE_N; ++k) {
                        const int i = lane * 4 + 256 * k;
                        if (i < lvl_n) *reinterpret_cast<fbbev_v4f*>(off_w + i) = pre[k];
                    }
                } else if ((lvl_n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {      // uniform
                    // round 6: the plane's 16-byte pieces are REQUESTED PRE_N at a time before the first is stored (one load, wait, LDS
                    // store per iteration before: up to 7 dependent round trips per camera and staged level)
                    for (int i0 = lane * 4; i0 < lvl_n; i0 += 256 * PRE_N) {
                        fbbev_v4f t[PRE_N];
#pragma unroll
                        for (int k = 0; k < PRE_N; ++k) { const int i = i0 + 256 * k; t[k] = *reinterpret_cast<const fbbev_v4f*>(src + (i < lvl_n ? i : i0)); }
#pragma unroll
                        for (int k = 0; k < PRE_N; ++k) { const int i = i0 + 256 * k; if (i < lvl_n) *reinterpret_cast<fbbev_v4f*>(off_w + i) = t[k]; }
                    }
                } else {                                                                     // DH is even: whole 4-byte words
                    for (int i = lane; i < lvl_n; i += 64) off_w[i] = src[i];
                }
                fbbev_wave_sync();
                const char* lplane = reinterpret_cast<const char*>(off_w);
                auto start = [&](int p, fbbev_daf_pending<DH, ET>& slot) {
                    const int z = p % ZA;
                    const float ox = o4[p >> 1][2 * (p & 1)], oy = o4[p >> 1][2 * (p & 1) + 1];
                    const float loc_w = rx[z] + __fdiv_rn(ox, fsw), loc_h = ry[z] + __fdiv_rn(oy, fsh);
                    const float weight = lg[0][p] * dw[z];
                    fbbev_daf_issue<DH, true, ET>(lplane, 0, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, sh, sw, weight, hit && !(diag & 2), slot);
                };
#pragma unroll
                for (int p = 0; p < NP - 1; ++p) start(p, pend[p]);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (p + NP - 1 < P) start(p + NP - 1, pend[(p + NP - 1) % NP]);
                    fbbev_sched_fence();
                    fbbev_daf_consume<DH, ET>(pend[p % NP], acc);
                    fbbev_sched_fence();
                }
                fbbev_wave_sync();                                                          // sampled before the next copy / projection overwrites it
                continue;
            }
            auto start = [&](int p, fbbev_daf_pending<DH, ET>& slot) {
                const int z = p % ZA;
                const float ox = o4[p >> 1][2 * (p & 1)], oy = o4[p >> 1][2 * (p & 1) + 1];
                const float loc_w = rx[z] + __fdiv_rn(ox, fsw), loc_h = ry[z] + __fdiv_rn(oy, fsh);
                const float weight = lg[0][p] * dw[z];
                fbbev_daf_issue<DH, false, ET>(plane, lvl_off, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, sh, sw, weight, hit && !(diag & 2), slot);
            };
#pragma unroll
            for (int p = 0; p < NP - 1; ++p) start(p, pend[p]);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (p + NP - 1 < P) start(p + NP - 1, pend[(p + NP - 1) % NP]);
                fbbev_sched_fence();
                fbbev_daf_consume<DH, ET>(pend[p % NP], acc);
                fbbev_sched_fence();
            }
        }
#pragma unroll
        for (int k = 0; k + 1 < FBBEV_DAF_MAXL; ++k)                   // the next level's weights move to row 0
#pragma unroll
            for (int p = 0; p < P; ++p) lg[k][p] = lg[k + 1][p];
    }
    const float inv = (float)(count > 1 ? count : 1);
    if constexpr (OP) {
        // the block's tail in the workgroup (see fbbev_daf_outproj): slots -> output_proj + residual + LayerNorm, rows to `slots`
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) { acc[c][0] = acc[c][0] / inv; acc[c][1] = acc[c][1] / inv; }
        __syncthreads();                                                // every wave is done with the query fragments
        fbbev_daf_put_channels<DH, KS>(xf, m, lane, acc);
        __syncthreads();
        if (wv < 4) {
            const int ql = 16 * wv + (lane & 15);
            const int ry = y0 + (ql >> 3), rx = x0 + (ql & 7);
            const bool live = ry < bev_h && rx < bev_w;
            fbbev_daf_outproj_ln<KS, E / 16>(op, xf, wv, lane, (long long)b * Q + (live ? (long long)ry * bev_w + rx : 0), live, slots, E);
        }
    } else {
        if (!valid) return;
        if ((diag >> 4) == 3 && acc[0][0] != 123.f) return;
        float* dst = slots + bq * E + m * DH;
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) {
            fbbev_v2f r;
            r[0] = acc[c][0] / inv; r[1] = acc[c][1] / inv;
            fbbev_st(reinterpret_cast<fbbev_v2f*>(dst + 2 * c), r);
        }
    }
}

// value_proj rows (B*Ncam*S, M*DH) -> head planes (B*Ncam, M, S, DH): standalone re-layout for callers that hold row-major
// tokens (tests; the product's projection writes planes directly, k_rows_linear_x3 `plane_S`)
__global__ void __launch_bounds__(256)
k_rows_to_head_planes(const float* __restrict__ rows, long long n_rows, int S, int M, int DH, float* __restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = n_rows * M * DH;
    if (i >= n) return;
    const int c = (int)(i % DH);
    const long long t = i / DH;
    const int m = (int)(t % M);
    const long long r = t / M;
    const long long bn = r / S, tok = r - bn * S;
    planes[((bn * M + m) * S + tok) * DH + c] = rows[i];
}

// ---------------------------------------------------------------- BEV self-attention, inference, one kernel (round 4)
// mmcv MultiScaleDeformableAttention.forward as the encoder layer uses it (bevformer_encoder.py:327-341; one level = the BEV grid
// itself, 4 points, value = the query tokens): the same construction as k_da_cross_attn_fused -- value_proj writes head planes
// (B, M, S, DH); a workgroup = the MH heads of an 8 x 8 patch of BEV queries, a wave = one head; the sampling_offsets (80 -> 64)
// and attention_weights (80 -> 32) projections run on the split-operand bf16 MFMA from the patch's query rows (+ positional
// rows) in LDS; softmax over the head's 4 logits in registers; `loc = ref + offset / (W, H)` and the bilinear samples as in
// k_msda_fwd_unit.  Replaces two k_rows_linear_x3 launches, the softmax launch and k_msda_fwd_unit, and the (B,Q,8,1,4,2) /
// (B,Q,8,1,4) tensors between them.  One level, P = 4 (mmcv's defaults, the only form the FB-OCC configs use).
// Rows of the weight matrices in the MODULE's order: offsets ((m*P + p)*2 + xy), logits (m*P + p).
#define FBBEV_MSF_P 4
__host__ __device__ inline size_t fbbev_msf_lds_bytes(int E, int M) { return fbbev_daf_xf_bytes(E) + (size_t)M * 64 * FBBEV_DAF_OS * 4; }

template <int DH, int MH, bool OP>
__global__ void __launch_bounds__(64 * MH, 4)      // 4 waves per SIMD = two workgroups per CU (65 KB of LDS each): 128 registers
k_msda_self_fused(const float* __restrict__ planes, const float* __restrict__ ref, const float* __restrict__ query, long long ldq,
                  const float* __restrict__ addend, long long ld_add, long long add_period,
                  const unsigned short* __restrict__ so_frag, const float* __restrict__ so_bias,
                  const unsigned short* __restrict__ aw_frag, const float* __restrict__ aw_bias,
                  int B, int Q, int bev_w, int S, int H, int W, float* __restrict__ out, fbbev_daf_outproj op) {
    constexpr int E = MH * DH, KS = (E + 31) / 32, P = FBBEV_MSF_P, NT = 64 * MH;
    static_assert(E % 16 == 0, "output_proj epilogue: whole 16-output tiles");
    static_assert(MH == 8, "a 16-output tile = the offsets of two heads / the logits of four");
    unsigned char* lds = reinterpret_cast<unsigned char*>(fbbev_dyn_lds_f32());
    unsigned short* xf = reinterpret_cast<unsigned short*>(lds);
    float* tiles = reinterpret_cast<float*>(lds + fbbev_daf_xf_bytes(E));                   // [MH][64][OS]
    const int bev_h = Q / bev_w;
    const int pxn = (bev_w + 7) / 8, pyn = (bev_h + 7) / 8;
    const long long n_wg = (long long)B * pxn * pyn, per_xcd = (n_wg + 7) / 8;
    const long long wgid = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wgid >= n_wg) return;
    const int b = (int)(wgid / ((long long)pxn * pyn)), pi = (int)(wgid - (long long)b * pxn * pyn);
    const int py = pi / pxn, px = pi - py * pxn;
    const int x0 = px * 8, y0 = py * 8;
    for (int i = threadIdx.x; i < 4 * KS * 64; i += NT) {                                   // (the batched form, fbbev_daf_query_fragments, costs this
        const int rt = i / (KS * 64), s = (i >> 6) % KS, ln = i & 63;                       //  128-register kernel two spilled values and 2 us)
        const int g = ln >> 4, j = ln & 15, ql = 16 * rt + j, c = 32 * s + 8 * g;
        const int qy = y0 + (ql >> 3), qx = x0 + (ql & 7);
        fbbev_v4f lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
        if (c < E && qy < bev_h && qx < bev_w) {
            const long long row = (long long)b * Q + (long long)qy * bev_w + qx;
            const float* src = query + row * ldq + c;
            lo4 = *reinterpret_cast<const fbbev_v4f*>(src); hi4 = *reinterpret_cast<const fbbev_v4f*>(src + 4);
            if (addend) {
                const float* a = addend + (row % add_period) * ld_add + c;
                lo4 = lo4 + *reinterpret_cast<const fbbev_v4f*>(a); hi4 = hi4 + *reinterpret_cast<const fbbev_v4f*>(a + 4);
            }
        }
        fbbev_bf16x8 h8, l8;
        fbbev_split_bf16x8(lo4, hi4, h8, l8);
        __builtin_memcpy(xf + (((rt * KS + s) * 2 + 0) * 64 + ln) * 8, &h8, 16);
        __builtin_memcpy(xf + (((rt * KS + s) * 2 + 1) * 64 + ln) * 8, &l8, 16);
    }
    __syncthreads();
    const int m = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    float* tile = tiles + (size_t)m * 64 * FBBEV_DAF_OS;
    const int qy = y0 + (lane >> 3), qx = x0 + (lane & 7);
    const bool valid = qy < bev_h && qx < bev_w;
    const long long bq = (long long)b * Q + (valid ? (long long)qy * bev_w + qx : 0);
    const fbbev_v2f rxy = *reinterpret_cast<const fbbev_v2f*>(ref + bq * 2);                // requested before the projections
    fbbev_v4f pacc[4];
    {   // offsets: tile m / 2 holds heads (m & ~1, m | 1); this head's 8 outputs are 4g + r with g / 2 == m & 1
        const int T = m >> 1;
        fbbev_daf_project<KS>(so_frag, T, xf, lane, pacc);
        if ((g >> 1) == (m & 1)) {
            fbbev_v4f bias4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = so_bias[16 * T + 4 * g + r];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const fbbev_v4f v = pacc[rt] + bias4;
                __builtin_memcpy(tile + (16 * rt + j) * FBBEV_DAF_OS + 4 * (g & 1), &v, 16);
            }
        }
    }
    {   // logits: tile m / 4 holds four heads; this head's P = 4 outputs are 4g + r with g == m & 3
        const int T = m >> 2;
        fbbev_daf_project<KS>(aw_frag, T, xf, lane, pacc);
        if (g == (m & 3)) {
            fbbev_v4f bias4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = aw_bias[16 * T + 4 * g + r];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const fbbev_v4f v = pacc[rt] + bias4;
                __builtin_memcpy(tile + (16 * rt + j) * FBBEV_DAF_OS + 8, &v, 16);
            }
        }
    }
    fbbev_wave_sync();
    fbbev_v4f o4[2], lg;
    __builtin_memcpy(&o4[0], tile + lane * FBBEV_DAF_OS, 16);
    __builtin_memcpy(&o4[1], tile + lane * FBBEV_DAF_OS + 4, 16);
    __builtin_memcpy(&lg, tile + lane * FBBEV_DAF_OS + 8, 16);
    float wts[P];
    {
        const float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) { wts[p] = __expf(lg[p] - mx); sum += wts[p]; }
        const float inv_sum = 1.f / sum;
#pragma unroll
        for (int p = 0; p < P; ++p) wts[p] *= inv_sum;
    }
    const float fsh = (float)H, fsw = (float)W;
    const char* plane = reinterpret_cast<const char*>(planes) + ((long long)b * MH + m) * (long long)S * DH * 4;
    fbbev_v2f acc[DH / 2];
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) { acc[c][0] = 0.f; acc[c][1] = 0.f; }
    fbbev_daf_pending<DH> pend[2];
    auto start = [&](int p, fbbev_daf_pending<DH>& slot) {
        const float ox = o4[p >> 1][2 * (p & 1)], oy = o4[p >> 1][2 * (p & 1) + 1];
        const float loc_w = rxy[0] + __fdiv_rn(ox, fsw), loc_h = rxy[1] + __fdiv_rn(oy, fsh);
        fbbev_daf_issue<DH>(plane, 0, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, H, W, wts[p], valid, slot);
    };
    start(0, pend[0]);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (p + 1 < P) start(p + 1, pend[(p + 1) & 1]);
        fbbev_sched_fence();
        fbbev_daf_consume<DH>(pend[p & 1], acc);
        fbbev_sched_fence();
    }
    if constexpr (OP) {                                                 // output_proj + residual + LayerNorm in the workgroup (its own instantiation)
        __syncthreads();                                                // every wave is done with the query fragments
        fbbev_daf_put_channels<DH, KS>(xf, m, lane, acc);
        __syncthreads();
        if (m < 4) {
            const int ql = 16 * m + (lane & 15);
            const int ry = y0 + (ql >> 3), rx = x0 + (ql & 7);
            const bool live = ry < bev_h && rx < bev_w;
            fbbev_daf_outproj_ln<KS, E / 16>(op, xf, m, lane, (long long)b * Q + (live ? (long long)ry * bev_w + rx : 0), live, out, E);
        }
    } else {
        if (!valid) return;
        float* dst = out + bq * E + m * DH;
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) fbbev_st(reinterpret_cast<fbbev_v2f*>(dst + 2 * c), acc[c]);
    }
}
