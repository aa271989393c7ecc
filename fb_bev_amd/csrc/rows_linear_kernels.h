// rows_linear_kernels.h -- y = x W^T + b (+ ReLU) for the (B*Q, C) row tensors of the backward projection, on the bf16 MFMA with
// SPLIT operands (round 3; the arithmetic of history_conv_x3_kernels.h: v = hi + lo in bf16, three MFMAs per product, fp32
// accumulate -- ~1e-5 relative, i.e. fp32-grade, against 16x the matrix rate of the fp32 MFMA).
//
// Why: at BASELINE configs[2] the encoder applies ~10 linear layers (80 -> 64 ... 512, 512 -> 80) to 160 000 query rows: 0.75 ms
// of the 2.44 ms forward+backward projection in the vendor library's fp32 GEMMs (87 TFLOP/s on the widest), where the rows
// themselves (51 MB in, 41-328 MB out) need 20-80 us at HBM speed.  With 3 bf16 MFMAs per product the kernels are bound by
// their rows, and bias + ReLU ride in the store epilogue (the FFN's ReLU was a separate 328 MB pass).
//
// Shape: workgroup = 4 waves x NT 16-row tiles (128 rows at NT = 2) x one 128-wide chunk of the outputs; K is walked in chunks of
// 128 channels (4 k-steps of 32) whose weight fragments -- hi and lo, prepared once per weight version by
// k_rows_linear_x3_fragments in the order [out chunk][K chunk][out tile][hi | lo][k-step][lane][8] -- are staged through LDS
// (8 KB per 16 outputs; 64 KB for a full chunk: two workgroups per CU) and shared by the 4 waves; the rows are read straight
// into registers (a lane = one row's 8 consecutive channels of a k-step: two float4), split there, never staged.  Optional
// `addend` (rows repeating with a period: the positional encoding of the BEV queries) is added to the row piece in fp32 first --
// the `query + query_pos` pass of the attention modules folded into the projections that consume it.
#pragma once
#include "rt.h"
#include "history_conv_x3_kernels.h"

#define FBBEV_RL_TILE_ELEMS (2 * 4 * 64 * 8)              // bf16 elements of one 16-output tile of one K chunk: [hi|lo][4 k-steps][lane][8]

// global -> LDS copy of n 16-byte pieces by a 256-thread workgroup, U pieces per thread REQUESTED before the first is stored: one
// round trip per batch (round 5: written as `for (i = tid; i < n; i += 256) dst[i] = src[i]` the compiler keeps a loop of load,
// s_waitcnt vmcnt(0), store -- n / 256 round trips in a row, 16 for a full 128-output chunk; profiles/r05_exp_weight_staging.md)
template <int U>
__device__ __forceinline__ void fbbev_stage_v4u(fbbev_v4u* __restrict__ dst, const fbbev_v4u* __restrict__ src, int n) {
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * U) {
        fbbev_v4u t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + 256 * u; t[u] = src[i < n ? i : i0]; }     // (clamped: unconditional loads)
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + 256 * u; if (i < n) dst[i] = t[u]; }
    }
}

// EPI = 1 (round 6, the training path's instantiation -- its own, so that the inference kernels keep their register budgets):
// out = ((x W^T + b) [ReLU]) * [mask > 0] + res, with `mask` (rows, O) the saved forward activation of a ReLU whose backward this
// product is (`threshold_backward` folded into the dgrad's store) and `res` (rows, O) a residual / running gradient sum; `res == out`
// is allowed (an element is read, then written, by one thread, and the stored value depends on the loaded one) -- the element-wise
// passes between the GEMMs of a backward
template <int NT, bool LN, int EPI = 0>
__global__ void __launch_bounds__(256, NT == 1 ? 3 : 2)
k_rows_linear_x3(const float* __restrict__ x, long long ldx, const unsigned short* __restrict__ wf, const float* __restrict__ bias,
                 float* __restrict__ out, long long ldo, long long rows, int I, int O, int relu, int n_kc, int n_oc, int RT,
                 const float* __restrict__ addend, long long ld_add, long long add_period, int plane_S, int plane_TS,
                 const float* __restrict__ res, long long ld_res, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float ln_eps,
                 const float* __restrict__ mask = nullptr, long long ld_mask = 0) {
    unsigned short* wl = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());          // [nmt][FBBEV_RL_TILE_ELEMS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int oc = (int)(blockIdx.x % n_oc);              // the output chunks of one row tile are neighbours: its rows stay in L2
    const long long rt0 = (long long)(blockIdx.x / n_oc) * RT;
    const int o0 = oc * 128;
    const int nmt = (O - o0 >= 128) ? 8 : (O - o0 + 15) / 16;                             // 16-output tiles of this chunk
    const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
    // RT consecutive 128-row tiles per workgroup when the whole K fits one chunk (n_kc == 1: the fragments are staged ONCE and
    // reused -- with one row tile per workgroup the 64 KB of fragments per 64 KB of output were half the L2 traffic of the
    // 80 -> 512 layer); RT = 1 otherwise
    for (int ri = 0; ri < RT; ++ri) {
        const long long r0 = ((rt0 + ri) * 4 + wave) * (16 * NT);
        if ((rt0 + ri) * 4 * 16 * NT >= rows) break;                                      // uniform
        fbbev_v4f acc[8][NT];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        long long ra[NT];                                                                 // row of the addend (x + addend[row % period])
#pragma unroll
        for (int t = 0; t < NT; ++t) ra[t] = addend ? (r0 + 16 * t + j) % add_period : 0;
        for (int kc = 0; kc < n_kc; ++kc) {
            const int c0 = kc * 128;
            // the chunk's row pieces first (raw, into registers): they are in flight while the fragments are staged
            fbbev_v4f raw[4][NT][2];
            bool okp[4][NT];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = c0 + 32 * s + 8 * g;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const long long r = r0 + 16 * t + j;
                    okp[s][t] = r < rows && c < I;                                        // I % 8 == 0: a piece is all in or all out
                    const float* p = x + (okp[s][t] ? r * ldx + c : 0);
                    raw[s][t][0] = *reinterpret_cast<const fbbev_v4f*>(p);
                    raw[s][t][1] = *reinterpret_cast<const fbbev_v4f*>(p + 4);
                    if (addend) {                                                         // uniform: x + addend[row % period] (query + query_pos)
                        const float* q = addend + (okp[s][t] ? ra[t] * ld_add + c : 0);
                        raw[s][t][0] = raw[s][t][0] + *reinterpret_cast<const fbbev_v4f*>(q);
                        raw[s][t][1] = raw[s][t][1] + *reinterpret_cast<const fbbev_v4f*>(q + 4);
                    }
                }
            }
            if (n_kc > 1 || ri == 0) {
                if (kc || ri) __syncthreads();                                            // the previous chunk's fragments are done with
                const fbbev_v4u* src = reinterpret_cast<const fbbev_v4u*>(wf + ((long long)oc * n_kc + kc) * 8 * FBBEV_RL_TILE_ELEMS);
                fbbev_stage_v4u<(NT == 1 ? 2 : 4)>(reinterpret_cast<fbbev_v4u*>(wl), src, nmt * (FBBEV_RL_TILE_ELEMS / 8));
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (c0 + 32 * s >= I) break;                                              // uniform: no k-step beyond the input width
                fbbev_bf16x8 xh[NT], xl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    fbbev_split_bf16x8(raw[s][t][0], raw[s][t][1], xh[t], xl[t]);
                    xh[t] = okp[s][t] ? xh[t] : zero8;
                    xl[t] = okp[s][t] ? xl[t] : zero8;
                }
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    if (mt >= nmt) break;                                                 // uniform
                    const fbbev_bf16x8 ah = fbbev_ld_bf16x8(wl + mt * FBBEV_RL_TILE_ELEMS + (s * 64 + lane) * 8);
                    const fbbev_bf16x8 al = fbbev_ld_bf16x8(wl + mt * FBBEV_RL_TILE_ELEMS + FBBEV_RL_TILE_ELEMS / 2 + (s * 64 + lane) * 8);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(al, xh[t], acc[mt][t]);
                        acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xl[t], acc[mt][t]);
                        acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xh[t], acc[mt][t]);
                    }
                }
            }
        }
        if constexpr (LN) {
            // LayerNorm epilogue (its own instantiation: in the common kernel its 8 extra vectors per row tile cost the 240-register
            // budget -- 336 registers = one wave per SIMD for EVERY launch).  n_oc == 1: the workgroup holds whole output rows): out = LN(x W^T + b + res) -- the `output_proj
            // + residual + norm` tail of an attention block / the FFN (bevformer_encoder.py:250-377) in the GEMM's store epilogue
            // instead of a k_layernorm_rows launch that re-reads the rows.  Two-pass statistics as k_layernorm_rows (mean, then
            // the biased variance of the deviations); a row's O values live in the 4 lanes (g = 0..3, same j) of its row tile:
            // two xor-shuffles per reduction.
            // (round 5: the bias / residual / LayerNorm pieces of a row tile are REQUESTED together, unconditionally, at clamped
            // addresses -- under their `if`s every piece was its own load, s_waitcnt vmcnt(0), use: up to 32 round trips in a row)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                const bool live = r < rows;
                fbbev_v4f v[8], pb[8], pr[8];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int o = 16 * mt + 4 * g, oc_ = (mt < nmt && o < O) ? o : 0;
                    pb[mt] = bias ? *reinterpret_cast<const fbbev_v4f*>(bias + oc_) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                    pr[mt] = res ? *reinterpret_cast<const fbbev_v4f*>(res + (live ? r : 0) * ld_res + oc_) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                }
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int o = 16 * mt + 4 * g;
                    const bool ok = mt < nmt && o < O;
                    v[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                    if (ok) {
                        v[mt] = acc[mt][t];
                        if (bias) v[mt] = v[mt] + pb[mt];
                        if (res && live) v[mt] = v[mt] + pr[mt];
                        s += (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
                    }
                }
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                const float mean = s / (float)O;
                float q = 0.f;
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int o = 16 * mt + 4 * g;
                    if (mt < nmt && o < O) {
                        v[mt] = v[mt] - fbbev_v4f{mean, mean, mean, mean};
                        q += (v[mt][0] * v[mt][0] + v[mt][1] * v[mt][1]) + (v[mt][2] * v[mt][2] + v[mt][3] * v[mt][3]);
                    }
                }
                q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
                const float inv = 1.0f / sqrtf(q / (float)O + ln_eps);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {                                           // (pb / pr are free: the LayerNorm pieces take their place)
                    const int o = 16 * mt + 4 * g, oc_ = (mt < nmt && o < O) ? o : 0;
                    pb[mt] = *reinterpret_cast<const fbbev_v4f*>(ln_w + oc_);
                    pr[mt] = *reinterpret_cast<const fbbev_v4f*>(ln_b + oc_);
                }
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int o = 16 * mt + 4 * g;
                    if (mt < nmt && o < O && live) {
                        const fbbev_v4f w4 = pb[mt], b4 = pr[mt];
                        fbbev_v4f y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = v[mt][e] * inv * w4[e] + b4[e];
                        fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), y);
                    }
                }
            }
            continue;
        }
        // accumulator register r of tile (mt, t) = output 16 mt + 4 g + r of row j: four consecutive outputs, one 16-byte store
        // head-plane output: a row's part of the element index ((bn * M) * S + token) * TS, the stride between two heads S * TS, 2^32 / TS
        long long row_at[NT];
        long long head_stride = 0;
        unsigned int ts_rcp = 0;
        if (plane_S > 0) {                                                                // uniform
            const int TS = plane_TS & 0xffff;
            const int Mh = O / TS;
            head_stride = (long long)plane_S * TS;
            ts_rcp = 0xffffffffu / (unsigned int)TS + 1u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                const long long bn = r / plane_S, tok = r - bn * plane_S;
                row_at[t] = (bn * Mh * plane_S + tok) * TS;
            }
        }
        if constexpr (EPI == 1) {
            // training epilogue: the residual / mask pieces of four output tiles are requested together (clamped, unconditional)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                const bool live = r < rows;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    fbbev_v4f pr[4], pm[4], pb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mt = 4 * half + q, o = o0 + 16 * mt + 4 * g;
                        const bool ok = live && mt < nmt && o < O;
                        pb[q] = bias ? *reinterpret_cast<const fbbev_v4f*>(bias + ((mt < nmt && o < O) ? o : 0)) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                        pr[q] = res ? *reinterpret_cast<const fbbev_v4f*>(res + (ok ? r * ld_res + o : 0)) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                        pm[q] = mask ? *reinterpret_cast<const fbbev_v4f*>(mask + (ok ? r * ld_mask + o : 0)) : fbbev_v4f{1.f, 1.f, 1.f, 1.f};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mt = 4 * half + q, o = o0 + 16 * mt + 4 * g;
                        if (!(live && mt < nmt && o < O)) continue;
                        fbbev_v4f v = acc[mt][t] + pb[q];
                        if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (pm[q][e] > 0.f ? v[e] : 0.f) + pr[q][e];
                        fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), v);
                    }
                }
            }
            continue;
        }
        fbbev_v4f pbias[NT >= 2 ? 8 : 1];                                                 // (requested together: see the LayerNorm epilogue;
        if constexpr (NT >= 2) {                                                          //  the 168-register NT = 1 build has no room for them)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int o = o0 + 16 * mt + 4 * g;
                pbias[mt] = bias ? *reinterpret_cast<const fbbev_v4f*>(bias + ((mt < nmt && o < O) ? o : 0)) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long long r = r0 + 16 * t + j;
            if (r >= rows) continue;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int o = o0 + 16 * mt + 4 * g;
                if (mt >= nmt || o >= O) continue;                                        // O % 4 == 0: a group is all in or all out
                fbbev_v4f v = acc[mt][t];
                if constexpr (NT >= 2) { if (bias) v = v + pbias[mt]; }
                else if (bias) v = v + *reinterpret_cast<const fbbev_v4f*>(bias + o);
                if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                if (plane_S > 0) {
                    // head-plane output (da_fused_kernels.h): rows are (sample-camera bn, token) pairs, outputs (head, channel);
                    // element (r, o) goes to out[((bn * M + o / TS) * S + token) * TS + o % TS], M = O / TS.  TS is even, so a
                    // channel pair never straddles two heads: two 8-byte stores.  Bits 16.. of plane_TS (round 5): 16-bit planes
                    // (1 bf16, 2 fp16; one nearest-even rounding of the fp32 result) -- a channel pair is one 4-byte store
                    // (round 5: the divisions by TS go through its reciprocal -- exact for o < 2^16 -- and the row's part of the address is
                    // formed once per row: as written first, ~50 integer instructions per 8-byte store made this epilogue longer than the GEMM)
                    const int TS = plane_TS & 0xffff, pet = plane_TS >> 16;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const unsigned int hd = fbbev_umulhi((unsigned int)(o + e), ts_rcp), ch = (unsigned int)(o + e) - hd * (unsigned int)TS;
                        const long long at = row_at[t] + (long long)hd * head_stride + ch;
                        if (pet) {
                            const unsigned int pk = pet == 1 ? fbbev_cvt_pk16<1>(v[e], v[e + 1]) : fbbev_cvt_pk16<2>(v[e], v[e + 1]);
                            fbbev_st(reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned short*>(out) + at), pk);
                        } else {
                            fbbev_v2f pr;
                            pr[0] = v[e]; pr[1] = v[e + 1];
                            fbbev_st(reinterpret_cast<fbbev_v2f*>(out + at), pr);
                        }
                    }
                    continue;
                }
                fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), v);
            }
        }
    }
}

// ---------------------------------------------------------------- the same product, PERSISTENT over row tiles (round 6)
// k_rows_linear_x3 runs at ~3 TB/s of rows in + out at every shape and every row-tile knob (profiles/r06_exp_rows_linear.md), where a
// copy with the same lane -> address pattern reaches 6 (tools/micro/row_access_bench.hip): a workgroup's tile is one serial chain --
// row pieces, fragment staging (two round trips + barrier), 90-144 MFMAs each behind its own LDS read, bias round trip, stores --
// with a second workgroup per CU as the only overlap.  Here, for one K chunk (I <= 128) and plain / head-plane output:
//   * a workgroup stays on its output chunk and walks every n_slots-th row tile: fragments and bias are staged ONCE (bias in LDS);
//   * the NEXT tile's row pieces are requested right after the current ones were split: in flight under the MFMAs and the stores;
//   * the fragment reads of an output tile are issued under the MFMAs of the tile before it (issue-order hints);
//   * NMT (output tiles of a chunk) and KS (k-steps) are compile-time: the MFMA phase is straight-line code.
// Same fragments, same product / accumulation order per element as k_rows_linear_x3: identical bits.
template <int NMT, int KS, int EPI>
__global__ void __launch_bounds__(256, 2)
k_rows_linear_x3p(const float* __restrict__ x, long long ldx, const unsigned short* __restrict__ wf, const float* __restrict__ bias,
                  float* __restrict__ out, long long ldo, long long rows, int I, int O, int relu, int n_oc, int n_slots,
                  int plane_S, int plane_TS, const float* __restrict__ res, long long ld_res, const float* __restrict__ mask,
                  long long ld_mask) {
    constexpr int NT = 2;
    unsigned short* wl = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());          // [NMT][FBBEV_RL_TILE_ELEMS]
    float* bl = reinterpret_cast<float*>(wl + NMT * FBBEV_RL_TILE_ELEMS);                // [16 NMT]: the chunk's bias (zeros: none / beyond O)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int oc = (int)(blockIdx.x % n_oc);              // the output chunks of one row tile are neighbours: its rows stay in L2
    const long long slot = blockIdx.x / n_oc;
    const int o0 = oc * 128;
    const long long tiles = (rows + 127) / 128;
    if (slot >= tiles) return;                                                            // uniform
    const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
    fbbev_v4f raw[KS][NT][2];
    auto request = [&](long long rt) {                                                    // a tile's row pieces, raw (clamped, unconditional)
        const long long r0 = (rt * 4 + wave) * (16 * NT);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                const float* p = x + ((r < rows && c < I) ? r * ldx + c : 0);
                raw[s][t][0] = *reinterpret_cast<const fbbev_v4f*>(p);
                raw[s][t][1] = *reinterpret_cast<const fbbev_v4f*>(p + 4);
            }
        }
    };
    request(slot);
    {
        const fbbev_v4u* src = reinterpret_cast<const fbbev_v4u*>(wf + (long long)oc * 8 * FBBEV_RL_TILE_ELEMS);      // n_kc == 1
        fbbev_stage_v4u<4>(reinterpret_cast<fbbev_v4u*>(wl), src, NMT * (FBBEV_RL_TILE_ELEMS / 8));
        for (int i = threadIdx.x; i < 16 * NMT; i += 256) bl[i] = (bias && o0 + i < O) ? bias[o0 + i] : 0.f;
    }
    __syncthreads();
    // head-plane output (as k_rows_linear_x3): the stride between two heads S * TS, 2^32 / TS
    long long head_stride = 0;
    unsigned int ts_rcp = 0;
    const int TS = plane_TS & 0xffff, pet = plane_TS >> 16;
    if (plane_S > 0) {
        head_stride = (long long)plane_S * TS;
        ts_rcp = 0xffffffffu / (unsigned int)TS + 1u;
    }
    for (long long rt = slot; rt < tiles; rt += n_slots) {
        const long long r0 = (rt * 4 + wave) * (16 * NT);
        fbbev_bf16x8 xh[KS][NT], xl[KS][NT];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool ok = r0 + 16 * t + j < rows && 32 * s + 8 * g < I;             // I % 8 == 0: a piece is all in or all out
                fbbev_split_bf16x8(raw[s][t][0], raw[s][t][1], xh[s][t], xl[s][t]);
                xh[s][t] = ok ? xh[s][t] : zero8;
                xl[s][t] = ok ? xl[s][t] : zero8;
            }
        fbbev_sched_fence();
        request(rt + n_slots < tiles ? rt + n_slots : rt);                                // the next tile (the last one again: unused)
        fbbev_sched_fence();
        fbbev_v4f acc[NMT][NT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const fbbev_bf16x8 ah = fbbev_ld_bf16x8(wl + mt * FBBEV_RL_TILE_ELEMS + (s * 64 + lane) * 8);
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(wl + mt * FBBEV_RL_TILE_ELEMS + FBBEV_RL_TILE_ELEMS / 2 + (s * 64 + lane) * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(al, xh[s][t], acc[mt][t]);
                    acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xl[s][t], acc[mt][t]);
                    acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xh[s][t], acc[mt][t]);
                }
            }
        // issue order: the two fragment reads of output tile (s, mt + 1) under the six MFMAs of tile (s, mt)
        FBBEV_SCHED_LDS_READ(2);
#pragma unroll
        for (int i = 0; i < KS * NMT - 1; ++i) { FBBEV_SCHED_MFMA(3); FBBEV_SCHED_LDS_READ(1); FBBEV_SCHED_MFMA(3); FBBEV_SCHED_LDS_READ(1); }
        FBBEV_SCHED_MFMA(6);
        fbbev_sched_fence();
        // epilogue: accumulator register r of tile (mt, t) = output 16 mt + 4 g + r of row j
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long long r = r0 + 16 * t + j;
            const bool live = r < rows;
            long long row_at = 0;
            if (plane_S > 0) {
                const long long bn = r / plane_S, tok = r - bn * plane_S;
                row_at = (bn * (O / TS) * plane_S + tok) * TS;
            }
            if constexpr (EPI == 1) {
                // training epilogue: the residual / mask pieces of four output tiles are requested together (clamped, unconditional)
#pragma unroll
                for (int m0 = 0; m0 < NMT; m0 += 4) {
                    fbbev_v4f pr[4], pm[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mt = m0 + q < NMT ? m0 + q : NMT - 1, o = o0 + 16 * mt + 4 * g;
                        const bool ok = live && o < O;
                        pr[q] = res ? *reinterpret_cast<const fbbev_v4f*>(res + (ok ? r * ld_res + o : 0)) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                        pm[q] = mask ? *reinterpret_cast<const fbbev_v4f*>(mask + (ok ? r * ld_mask + o : 0)) : fbbev_v4f{1.f, 1.f, 1.f, 1.f};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (m0 + q >= NMT) continue;
                        const int mt = m0 + q < NMT ? m0 + q : NMT - 1, o = o0 + 16 * mt + 4 * g;
                        if (!(live && o < O)) continue;
                        fbbev_v4f v = acc[mt][t] + *reinterpret_cast<const fbbev_v4f*>(bl + 16 * mt + 4 * g);
                        if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (pm[q][e] > 0.f ? v[e] : 0.f) + pr[q][e];
                        fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), v);
                    }
                }
                continue;
            }
            if (!live) continue;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int o = o0 + 16 * mt + 4 * g;
                if (o >= O) continue;                                                     // O % 4 == 0: a group is all in or all out
                fbbev_v4f v = acc[mt][t];
                if (bias) v = v + *reinterpret_cast<const fbbev_v4f*>(bl + 16 * mt + 4 * g);   // (uniform; no `+ 0`: -0 stays -0 as in k_rows_linear_x3)
                if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                if (plane_S > 0) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const unsigned int hd = fbbev_umulhi((unsigned int)(o + e), ts_rcp), ch = (unsigned int)(o + e) - hd * (unsigned int)TS;
                        const long long at = row_at + (long long)hd * head_stride + ch;
                        if (pet) {
                            const unsigned int pk = pet == 1 ? fbbev_cvt_pk16<1>(v[e], v[e + 1]) : fbbev_cvt_pk16<2>(v[e], v[e + 1]);
                            fbbev_st(reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned short*>(out) + at), pk);
                        } else {
                            fbbev_v2f pr;
                            pr[0] = v[e]; pr[1] = v[e + 1];
                            fbbev_st(reinterpret_cast<fbbev_v2f*>(out + at), pr);
                        }
                    }
                    continue;
                }
                fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), v);
            }
        }
    }
}

// W (O, I) fp32 row-major -> split bf16 A fragments, order [out chunk][K chunk][out tile][hi | lo][k-step][lane][8]; element e of a
// lane = W[128 oc + 16 mt + lane % 16][128 kc + 32 s + 8 (lane / 16) + e], zero outside the matrix.
__global__ void __launch_bounds__(256)
k_rows_linear_x3_fragments(const float* __restrict__ w, int O, int I, int n_oc, int n_kc, unsigned short* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;               // one lane-fragment per thread
    const long long n = (long long)n_oc * n_kc * 8 * 4 * 64;
    if (i >= n) return;
    const int lane = (int)(i & 63), s = (int)((i >> 6) & 3), mt = (int)((i >> 8) & 7);
    const long long blk = i >> 11;
    const int kc = (int)(blk % n_kc), oc = (int)(blk / n_kc);
    const int o = 128 * oc + 16 * mt + (lane & 15);
    fbbev_v4f lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 128 * kc + 32 * s + 8 * (lane >> 4) + e;
        const float v = (o < O && c < I) ? w[(long long)o * I + c] : 0.f;
        if (e < 4) lo[e] = v; else hi[e - 4] = v;
    }
    fbbev_bf16x8 h8, l8;
    fbbev_split_bf16x8(lo, hi, h8, l8);
    unsigned short* base = dst + (blk * 8 + mt) * FBBEV_RL_TILE_ELEMS + (s * 64 + lane) * 8;
    __builtin_memcpy(base, &h8, 16);
    __builtin_memcpy(base + FBBEV_RL_TILE_ELEMS / 2, &l8, 16);
}

// ---------------------------------------------------------------- the FFN pair in one kernel (round 4)
// mmcv FFN of the encoder layer (bevformer_encoder.py:250-377, cfg ffn_cfgs: Linear(80 -> 320) + ReLU, Linear(320 -> 80), + identity,
// then the layer's LayerNorm): y = [LN](x + W2 relu(W1 x + b1) + b2).  As two k_rows_linear_x3 launches the 320-wide hidden rows
// were written and re-read (205 MB each way at 160 000 rows).  Here a wave keeps its 32 rows' x fragments and output accumulators
// in registers and walks the hidden units in chunks of 64: GEMM 1 (4 tiles x KS1 k-steps) -> + b1, ReLU -> split into bf16 hi / lo
// and dropped into a wave-private LDS buffer ALREADY in B-fragment order (the accumulator register of lane (g, j) for hidden unit
// 16 mt + 4 g + r of row j is element 4 (g & 1) + r of fragment lane (2 (mt & 1) + g / 2, j) of k-step mt / 2) -> GEMM 2 (MT2 tiles x
// 2 k-steps) accumulates into the outputs.  The chunk's weight fragments (W1: 24 KB, W2: 20 KB, split operands) are staged through
// LDS once per workgroup and chunk.  Same split-operand arithmetic as k_rows_linear_x3 (three MFMAs per product) for both GEMMs.
// LDS: 24 + 20 + 4 x 8 KB = 76 KB -> two workgroups per CU.
#define FBBEV_FFN_HC 64                                    // hidden units the launcher requires the width to be a multiple of
// HC = hidden units per chunk: 64 (4 tiles = 2 k-steps of GEMM 2, 76 KB of LDS: two workgroups per CU) or 32 (2 tiles = 1 k-step,
// 38 KB: three workgroups per CU at the kernel's 164 registers -- twice the barriers, 1.5x the waves to hide them behind)
template <int KS1, int MT2, int HC>
__host__ __device__ constexpr int fbbev_ffn_lds_bytes() { return ((HC / 16) * 2 * KS1 + MT2 * 2 * (HC / 32)) * 1024 + 4 * 2 * (HC / 32) * 2 * 1024; }

// The attention block's tail in FRONT of the FFN pair (round 5, `PRE`): y1 = LayerNorm0(x0 W0^T + b0 + res0) -- `output_proj` +
// residual + the layer's norm after the cross-attention, bevformer_encoder.py:250-377 -- computed by the same wave for its 32 rows,
// kept in registers as the FFN's residual and re-laid out through the wave's hidden-fragment buffer into the B fragments of GEMM 1.
// Replaces a k_rows_linear_x3<., true> launch and the write + re-read of the (rows, E) tensor between the two kernels.
struct fbbev_ffn_pre {
    const unsigned short* w0f;         // output_proj.weight (E, E) as split bf16 fragments (k_rows_linear_x3_fragments)
    const float* b0;                   // (E)
    const float* res0;                 // residual rows (rows, ld_res0), may be null
    long long ld_res0;
    const float* ln0_w;                // LayerNorm0 weight / bias (E)
    const float* ln0_b;
    float eps0;
};
// weight region = max(W1 chunk + W2 chunk, the MT2 x KS1 fragments of W0) KB + the four waves' hidden-fragment buffers
template <int KS1, int MT2, int HC>
__host__ __device__ constexpr int fbbev_ffn_pre_wregion_kb() {
    return ((HC / 16) * 2 * KS1 + MT2 * 2 * (HC / 32)) > MT2 * 2 * KS1 ? ((HC / 16) * 2 * KS1 + MT2 * 2 * (HC / 32)) : MT2 * 2 * KS1;
}
template <int KS1, int MT2, int HC>
__host__ __device__ constexpr int fbbev_ffn_pre_lds_bytes() { return fbbev_ffn_pre_wregion_kb<KS1, MT2, HC>() * 1024 + 4 * 2 * (HC / 32) * 2 * 1024; }

template <int KS1, int MT2, bool LN, int HC, bool PRE = false>
__global__ void __launch_bounds__(256, (HC == 32 && !PRE) ? 3 : 2)
k_rows_ffn_x3(const float* __restrict__ x, long long ldx, const unsigned short* __restrict__ w1f, const float* __restrict__ b1,
              const unsigned short* __restrict__ w2f, const float* __restrict__ b2, float* __restrict__ out, long long ldo,
              long long rows, int I, int H, int O, int n_kc2, const float* __restrict__ res, long long ld_res,
              const float* __restrict__ ln_w, const float* __restrict__ ln_b, float ln_eps, fbbev_ffn_pre pre) {
    constexpr int NT = 2, T1 = HC / 16, S2 = HC / 32;
    constexpr int WREGION = PRE ? fbbev_ffn_pre_wregion_kb<KS1, MT2, HC>() * 512 : (T1 * 2 * KS1 + MT2 * 2 * S2) * 512;   // bf16 elements
    unsigned short* w1s = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());          // [T1][hi|lo][KS1][64][8]
    unsigned short* w2s = w1s + T1 * 2 * KS1 * 512;                                        // [MT2][hi|lo][S2][64][8]
    unsigned short* hb = w1s + WREGION + (threadIdx.x >> 6) * (NT * S2 * 2 * 512);         // this wave's [t][s2][hi|lo][64][8]
    // b1 (H floats) behind the four waves' buffers (round 5; the launcher adds H * 4 bytes): a chunk's bias pieces come by ds_read --
    // as global loads they sat BEHIND the next chunk's prefetch in the in-order vmcnt and every GEMM 1 waited for that round trip
    float* b1s = reinterpret_cast<float*>(w1s + WREGION + 4 * (NT * S2 * 2 * 512));
    float b1r[4];                                                                           // (requested here, stored behind the rows' loads)
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = (int)threadIdx.x + 256 * u; b1r[u] = b1[i < H ? i : 0]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * (16 * NT);
    // this wave's rows as split B fragments, once
    fbbev_bf16x8 xh[KS1][NT], xl[KS1][NT];
    {
        const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const int c = 32 * s + 8 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                const bool ok = r < rows && c < I;
                const float* p = x + (ok ? r * ldx + c : 0);
                const fbbev_v4f a0 = *reinterpret_cast<const fbbev_v4f*>(p), a1 = *reinterpret_cast<const fbbev_v4f*>(p + 4);
                fbbev_split_bf16x8(a0, a1, xh[s][t], xl[s][t]);
                xh[s][t] = ok ? xh[s][t] : zero8;
                xl[s][t] = ok ? xl[s][t] : zero8;
            }
        }
    }
    // a chunk's weight fragments: W1 tiles T1 c .. T1 c + T1 - 1 (k-steps 0..KS1-1 of their single K chunk), W2 tiles 0..MT2-1 at hidden
    // units [HC c, HC c + HC) = k-steps ks2 .. of K chunk kc2; 16-byte pieces, [hi | lo] kept apart as in the fragment arrays.  Round 5:
    // `request` puts every piece of the chunk in flight (registers), `commit` stores them to LDS -- one round trip per chunk instead of
    // one per piece, and with PRE (two waves per SIMD: 256 registers) the NEXT chunk is requested before the current chunk's GEMMs
    constexpr int N1 = T1 * 2 * KS1 * 64, N2 = MT2 * 2 * S2 * 64, I1 = (N1 + 255) / 256, I2 = (N2 + 255) / 256;
    static_assert(N1 >= 256 && N2 >= 256, "clamped loads");
    constexpr bool PF = PRE && HC == 32;                                                    // (HC = 64 holds 44 registers of pieces: no room across its GEMMs)
    fbbev_v4u st1[I1], st2[I2];
    auto request = [&](int c) {
#pragma unroll
        for (int k = 0; k < I1; ++k) {
            const int i_ = (int)threadIdx.x + 256 * k, i = i_ < N1 ? i_ : (int)threadIdx.x;
            const int ln = i & 63, s = (i >> 6) % KS1, h = (i / (64 * KS1)) & 1, mt = i / (64 * KS1 * 2);
            st1[k] = *reinterpret_cast<const fbbev_v4u*>(w1f + (long long)(T1 * c + mt) * FBBEV_RL_TILE_ELEMS + h * (FBBEV_RL_TILE_ELEMS / 2) + (s * 64 + ln) * 8);
        }
        const int u0 = c * HC, kc2 = u0 >> 7, ks2 = (u0 & 127) >> 5;                        // K chunk / first k-step of the hidden chunk in W2's fragments
#pragma unroll
        for (int k = 0; k < I2; ++k) {
            const int i_ = (int)threadIdx.x + 256 * k, i = i_ < N2 ? i_ : (int)threadIdx.x;
            const int ln = i & 63, s = (i >> 6) % S2, h = (i / (64 * S2)) & 1, mt = i / (64 * S2 * 2);
            st2[k] = *reinterpret_cast<const fbbev_v4u*>(w2f + ((long long)kc2 * 8 + mt) * FBBEV_RL_TILE_ELEMS + h * (FBBEV_RL_TILE_ELEMS / 2) + ((ks2 + s) * 64 + ln) * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < I1; ++k) { const int i = (int)threadIdx.x + 256 * k; if (i < N1) reinterpret_cast<fbbev_v4u*>(w1s)[i] = st1[k]; }
#pragma unroll
        for (int k = 0; k < I2; ++k) { const int i = (int)threadIdx.x + 256 * k; if (i < N2) reinterpret_cast<fbbev_v4u*>(w2s)[i] = st2[k]; }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = (int)threadIdx.x + 256 * u; if (i < H) b1s[i] = b1r[u]; }
    for (int i = (int)threadIdx.x + 1024; i < H; i += 256) b1s[i] = b1[i];                  // (H > 1024: not an FB-OCC width); visible behind the
                                                                                            // __syncthreads that precedes the first GEMM 1
    fbbev_v4f y1[PRE ? MT2 : 1][NT];                                                        // PRE: LayerNorm0's output = the FFN's residual
    if constexpr (PRE) {
        // W0's fragments (tiles 0..MT2-1, k-steps 0..KS1-1 of its single K chunk) through the weight region, shared by the 4 waves
        {   // (all pieces requested before the first is stored: one round trip, see fbbev_stage_v4u)
            constexpr int N0 = MT2 * 2 * KS1 * 64, I0 = (N0 + 255) / 256;
            static_assert(N0 >= 256, "clamped loads");
            fbbev_v4u st0[I0];
#pragma unroll
            for (int k = 0; k < I0; ++k) {
                const int i_ = (int)threadIdx.x + 256 * k, i = i_ < N0 ? i_ : (int)threadIdx.x;
                const int ln = i & 63, s = (i >> 6) % KS1, h = (i / (64 * KS1)) & 1, mt = i / (64 * KS1 * 2);
                st0[k] = *reinterpret_cast<const fbbev_v4u*>(pre.w0f + (long long)mt * FBBEV_RL_TILE_ELEMS + h * (FBBEV_RL_TILE_ELEMS / 2) + (s * 64 + ln) * 8);
            }
#pragma unroll
            for (int k = 0; k < I0; ++k) {
                const int i = (int)threadIdx.x + 256 * k;
                if (i < N0) reinterpret_cast<fbbev_v4u*>(w1s)[i] = st0[k];
            }
        }
        __syncthreads();
        // bias and residual pieces of LayerNorm0's input: requested here, unconditionally (clamped), they arrive under GEMM 0
        fbbev_v4f pb0[MT2], pr0[MT2][NT];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int o = 16 * mt + 4 * g, oc_ = o < I ? o : 0;
            pb0[mt] = *reinterpret_cast<const fbbev_v4f*>(pre.b0 + oc_);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                pr0[mt][t] = pre.res0 ? *reinterpret_cast<const fbbev_v4f*>(pre.res0 + (r < rows ? r : 0) * pre.ld_res0 + oc_) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) y1[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const fbbev_bf16x8 ah = fbbev_ld_bf16x8(w1s + (((mt * 2 + 0) * KS1 + s) * 64 + lane) * 8);
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(w1s + (((mt * 2 + 1) * KS1 + s) * 64 + lane) * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    y1[mt][t] = fbbev_mfma_f32_16x16x32_bf16(al, xh[s][t], y1[mt][t]);
                    y1[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xl[s][t], y1[mt][t]);
                    y1[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, xh[s][t], y1[mt][t]);
                }
            }
        }
        // + b0 + res0 -> LayerNorm0 (two-pass statistics, as k_rows_linear_x3<., true>); the FFN's input width I == W0's output width
        fbbev_v4f pw0[MT2], pq0[MT2];                                                       // LayerNorm0's weight / bias pieces, requested together
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int o = 16 * mt + 4 * g, oc_ = o < I ? o : 0;
            pw0[mt] = *reinterpret_cast<const fbbev_v4f*>(pre.ln0_w + oc_);
            pq0[mt] = *reinterpret_cast<const fbbev_v4f*>(pre.ln0_b + oc_);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long long r = r0 + 16 * t + j;
            const bool live = r < rows;
            float sm = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const int o = 16 * mt + 4 * g;
                if (o < I) {
                    y1[mt][t] = y1[mt][t] + pb0[mt];
                    if (pre.res0 && live) y1[mt][t] = y1[mt][t] + pr0[mt][t];
                    sm += (y1[mt][t][0] + y1[mt][t][1]) + (y1[mt][t][2] + y1[mt][t][3]);
                } else {
                    y1[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
            sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
            const float mean = sm / (float)I;
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                if (16 * mt + 4 * g < I) {
                    y1[mt][t] = y1[mt][t] - fbbev_v4f{mean, mean, mean, mean};
                    q += (y1[mt][t][0] * y1[mt][t][0] + y1[mt][t][1] * y1[mt][t][1]) + (y1[mt][t][2] * y1[mt][t][2] + y1[mt][t][3] * y1[mt][t][3]);
                }
            }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float inv = 1.0f / sqrtf(q / (float)I + pre.eps0);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const int o = 16 * mt + 4 * g;
                if (o < I) {
                    const fbbev_v4f w4 = pw0[mt], b4 = pq0[mt];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y1[mt][t][e] = y1[mt][t][e] * inv * w4[e] + b4[e];
                }
            }
        }
        // accumulator layout -> B fragments of GEMM 1, one k-step (= two 16-output tiles) at a time through the wave's hidden buffer:
        // output 16 mt + 4 g + r of row j is element 4 (g & 1) + r of fragment lane (2 (mt & 1) + g / 2, j) of k-step mt / 2
        const fbbev_bf16x8 zero8 = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {
                const int mt = 2 * s + hm;
                if (mt >= MT2) break;
                const int gl = 2 * hm + (g >> 1), e0 = 4 * (g & 1);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    fbbev_bf16x8 h8, l8;
                    fbbev_split_bf16x8(y1[mt][t], fbbev_v4f{0.f, 0.f, 0.f, 0.f}, h8, l8);
                    __builtin_memcpy(hb + (((t * S2 * 2 + 0) * 64 + gl * 16 + j) * 8 + e0), &h8, 8);
                    __builtin_memcpy(hb + (((t * S2 * 2 + 1) * 64 + gl * 16 + j) * 8 + e0), &l8, 8);
                }
            }
            fbbev_wave_sync();
            const bool okc = 32 * s + 8 * g < I;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                xh[s][t] = fbbev_ld_bf16x8(hb + ((t * S2 * 2 + 0) * 64 + lane) * 8);
                xl[s][t] = fbbev_ld_bf16x8(hb + ((t * S2 * 2 + 1) * 64 + lane) * 8);
                xh[s][t] = okc ? xh[s][t] : zero8;
                xl[s][t] = okc ? xl[s][t] : zero8;
            }
            fbbev_wave_sync();                                                              // read before the next k-step overwrites
        }
    }
    if constexpr (PF) request(0);
    fbbev_v4f acc2[MT2][NT];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    const int n_chunks = H / HC;
    for (int c = 0; c < n_chunks; ++c) {
        if (!PF) request(c);                                                                // (PF: in flight since the previous chunk's GEMMs / the PRE block)
        if (c || PRE) __syncthreads();                                                      // the previous chunk's weights (PRE: W0's) are done with
        commit();
        if (PF && c + 1 < n_chunks) request(c + 1);                                         // (the pieces' registers are free again)
        __syncthreads();
        // GEMM 1 + bias + ReLU -> hidden fragments of this wave
#pragma unroll
        for (int mt = 0; mt < T1; ++mt) {
            fbbev_v4f acc1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc1[t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const fbbev_bf16x8 ah = fbbev_ld_bf16x8(w1s + (((mt * 2 + 0) * KS1 + s) * 64 + lane) * 8);
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(w1s + (((mt * 2 + 1) * KS1 + s) * 64 + lane) * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc1[t] = fbbev_mfma_f32_16x16x32_bf16(al, xh[s][t], acc1[t]);
                    acc1[t] = fbbev_mfma_f32_16x16x32_bf16(ah, xl[s][t], acc1[t]);
                    acc1[t] = fbbev_mfma_f32_16x16x32_bf16(ah, xh[s][t], acc1[t]);
                }
            }
            const fbbev_v4f bias4 = *reinterpret_cast<const fbbev_v4f*>(b1s + HC * c + 16 * mt + 4 * g);
            const int s2 = mt >> 1, gl = 2 * (mt & 1) + (g >> 1), e0 = 4 * (g & 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fbbev_v4f v = acc1[t] + bias4;
                v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                // hi = bf16(v), lo = bf16(v - hi): the split of fbbev_split_bf16x8 on four values
                fbbev_bf16x8 h8, l8;
                fbbev_split_bf16x8(v, fbbev_v4f{0.f, 0.f, 0.f, 0.f}, h8, l8);
                unsigned short* dh = hb + ((((t * S2 + s2) * 2 + 0) * 64 + gl * 16 + j) * 8 + e0);
                unsigned short* dl = hb + ((((t * S2 + s2) * 2 + 1) * 64 + gl * 16 + j) * 8 + e0);
                __builtin_memcpy(dh, &h8, 8);                                             // the first four elements
                __builtin_memcpy(dl, &l8, 8);
            }
        }
        fbbev_wave_sync();
        // GEMM 2: outputs += W2[:, chunk] . hidden
#pragma unroll
        for (int s2 = 0; s2 < S2; ++s2) {
            fbbev_bf16x8 hh[NT], hl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                hh[t] = fbbev_ld_bf16x8(hb + (((t * S2 + s2) * 2 + 0) * 64 + lane) * 8);
                hl[t] = fbbev_ld_bf16x8(hb + (((t * S2 + s2) * 2 + 1) * 64 + lane) * 8);
            }
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const fbbev_bf16x8 ah = fbbev_ld_bf16x8(w2s + (((mt * 2 + 0) * S2 + s2) * 64 + lane) * 8);
                const fbbev_bf16x8 al = fbbev_ld_bf16x8(w2s + (((mt * 2 + 1) * S2 + s2) * 64 + lane) * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc2[mt][t] = fbbev_mfma_f32_16x16x32_bf16(al, hh[t], acc2[mt][t]);
                    acc2[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, hl[t], acc2[mt][t]);
                    acc2[mt][t] = fbbev_mfma_f32_16x16x32_bf16(ah, hh[t], acc2[mt][t]);
                }
            }
        }
        fbbev_wave_sync();                                                                  // this wave's hidden fragments are read
    }
    // epilogue: + b2 (+ residual -> LayerNorm), as k_rows_linear_x3; its parameter / residual pieces requested together (clamped)
    fbbev_v4f pb2[MT2], pw[LN ? MT2 : 1], pq[LN ? MT2 : 1], pres[PRE ? 1 : MT2][NT];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
        const int o = 16 * mt + 4 * g, oc_ = o < O ? o : 0;
        pb2[mt] = *reinterpret_cast<const fbbev_v4f*>(b2 + oc_);
        if constexpr (LN) { pw[mt] = *reinterpret_cast<const fbbev_v4f*>(ln_w + oc_); pq[mt] = *reinterpret_cast<const fbbev_v4f*>(ln_b + oc_); }
        if constexpr (!PRE) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long long r = r0 + 16 * t + j;
                pres[mt][t] = res ? *reinterpret_cast<const fbbev_v4f*>(res + (r < rows ? r : 0) * ld_res + oc_) : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long r = r0 + 16 * t + j;
        const bool live = r < rows;
        fbbev_v4f v[MT2];
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int o = 16 * mt + 4 * g;
            v[mt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            if (o < O) {
                v[mt] = acc2[mt][t] + pb2[mt];
                if constexpr (PRE) v[mt] = v[mt] + y1[mt][t];                                // add_identity: the FFN's own input
                else if (res && live) v[mt] = v[mt] + pres[mt][t];
                s += (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
            }
        }
        if constexpr (LN) {
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            const float mean = s / (float)O;
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                if (16 * mt + 4 * g < O) {
                    v[mt] = v[mt] - fbbev_v4f{mean, mean, mean, mean};
                    q += (v[mt][0] * v[mt][0] + v[mt][1] * v[mt][1]) + (v[mt][2] * v[mt][2] + v[mt][3] * v[mt][3]);
                }
            }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float inv = 1.0f / sqrtf(q / (float)O + ln_eps);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const int o = 16 * mt + 4 * g;
                if (o < O) {
                    const fbbev_v4f w4 = pw[mt], b4 = pq[mt];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[mt][e] = v[mt][e] * inv * w4[e] + b4[e];
                }
            }
        }
        if (!live) continue;
        if (ldo < 0) {
            // planes (round 6): out is (rows / S, O, S) with S = -ldo rows per image -- the (B, C, Y, X) tensor the next consumer takes
            // (the re-add epilogue of the final pooling), written here instead of rows + a transposing pass.  The 16 lanes of a
            // row tile hold 16 consecutive rows of one feature: 64-byte runs.
            const unsigned int S = (unsigned int)(-ldo), ru = (unsigned int)r, bn = ru / S, tok = ru - bn * S;   // rows < 2^31: checked
            float* ob = out + (size_t)bn * O * S + tok;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const int o = 16 * mt + 4 * g;
                if (o < O) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fbbev_st(ob + (size_t)(o + e) * S, v[mt][e]);
                }
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int o = 16 * mt + 4 * g;
            if (o < O) fbbev_st(reinterpret_cast<fbbev_v4f*>(out + r * ldo + o), v[mt]);
        }
    }
}
