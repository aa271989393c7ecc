// msda_bwd_kernels.h -- MultiScaleDeformableAttention backward without floating-point global atomics (round 3).
//
// k_msda_bwd (msda_kernels.h) follows ms_deform_attn_cuda_kernel.cuh's col2im: every (sample, channel) adds its four
// corner products into grad_value with fp32 global atomics -- 205 M of them for the BEV self-attention of BASELINE
// configs[2] (B = 4, 40 000 queries, 8 heads, 4 points, Dh = 10), 1.48 ms, bound by the L2's read-modify-write rate, and the
// sum order (hence the low bits) changes from run to run.  Here, as in k_da_cross_attn_bwd_scatter, the value gradient is
// accumulated in LDS as 64-bit FIXED POINT (integer adds commute: bit-reproducible) -- but a self-attention plane is the
// whole 200 x 200 BEV, far beyond LDS, and the queries that touch a token are not known in advance.  So the token space is
// cut into BANDS of rows and the queries are binned by the rows they reach:
//   k_msda_row_ranges    (b, q, level) -> [first, last] token row any of the M*P samples of that query can touch
//                        (one row of slack either side: the scatter kernel must never find a corner this pass excluded);
//   k_msda_band_queries  (b, band) -> [qlo, qhi): the span of query indices whose range meets the band.  For raster-ordered
//                        BEV queries sampling around themselves that is the band's rows plus a halo; for arbitrary sampling
//                        it degrades to [0, Q) -- slower, never wrong;
//   k_msda_bwd_scatter   workgroup = (b, head, band): walks its query span (lane = query), re-evaluates the samples of the
//                        band's level and adds the corners that fall inside the band into the LDS plane; the band's tokens
//                        are then written to grad_value ONCE, by this workgroup alone (no partial planes, no reduction pass,
//                        no pre-zeroed grad_value);
//   k_msda_bwd<GW,false> the unit-owned gradients (sampling locations, attention weights) -- the old kernel minus its atomics.
// Fixed point: scale 2^(30 - ex) with max|grad_output| * max(1, max|attn|) < 2^ex over the workgroup's query span; every
// product is rounded once to that grid (relative 2^-30 of the largest term), the sum is exact, one rounding back to fp32.
#pragma once
#include "rt.h"
#include "msda_kernels.h"
#include "da_kernels.h"

struct fbbev_msda_band { int level, r0, r1, w, h, ls; };

// band index -> (level, rows [r0, r1)): level l is cut into ceil(h_l / rpb_l) bands of rpb_l = max(1, budget / w_l) rows.
// The host applies the same rule to count the bands (fbbev_msda_bwd_ws).
__device__ __forceinline__ fbbev_msda_band fbbev_msda_band_of(int band, int L, const int64_t* __restrict__ ss,
                                                              const int64_t* __restrict__ lsi, int budget) {
    fbbev_msda_band r = {-1, 0, 0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
        const int h = (int)ss[2 * l], w = (int)ss[2 * l + 1];
        int rpb = budget / (w > 0 ? w : 1);
        rpb = rpb < 1 ? 1 : rpb;
        const int nb = (h + rpb - 1) / rpb;
        if (band < nb) {
            r.level = l; r.r0 = band * rpb; r.r1 = (r.r0 + rpb < h) ? r.r0 + rpb : h; r.w = w; r.h = h; r.ls = (int)lsi[l];
            return r;
        }
        band -= nb;
    }
    return r;
}

// ranges[(b*L + l)*Q + q] = first | last << 16 (first > last: the query has no sample on that level)
__global__ void __launch_bounds__(256)
k_msda_row_ranges(long long n, const int64_t* __restrict__ ss, const float* __restrict__ loc, int M, int L, int Q, int P,
                  unsigned int* __restrict__ ranges) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        const long long bl = i / Q;
        const int l = (int)(bl % L);
        const long long b = bl / L;
        const int h = (int)ss[2 * l];
        int lo = 0x7fff, hi = -1;
        for (int m = 0; m < M; ++m) {
            const float* lp = loc + ((((b * Q + q) * M + m) * L + l) * (long long)P) * 2;
            for (int p0 = 0; p0 < P; p0 += 4) {                               // (round 6: four row coordinates requested together)
                float ly[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ly[e] = lp[2 * (p0 + e < P ? p0 + e : p0) + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h_im = ly[e] * (float)h - 0.5f;
                    if (p0 + e < P && h_im > -2.f && h_im < (float)h + 1.f) {   // wider than the sampler's own test: slack
                        const int r = (int)floorf(h_im);
                        lo = r - 1 < lo ? r - 1 : lo;
                        hi = r + 2 > hi ? r + 2 : hi;
                    }
                }
            }
        }
        lo = lo < 0 ? 0 : lo;
        hi = hi > h - 1 ? h - 1 : hi;
        ranges[i] = hi >= lo ? ((unsigned)lo | ((unsigned)hi << 16)) : 0x00007fffu;
    }
}

// qrange[(b*NB + band)*2 + {0, 1}] = [qlo, qhi)
__global__ void __launch_bounds__(256)
k_msda_band_queries(const int64_t* __restrict__ ss, const int64_t* __restrict__ lsi, const unsigned int* __restrict__ ranges,
                    int L, int Q, int NB, int budget, int* __restrict__ qrange) {
    __shared__ int red[2][4];
    const int band = blockIdx.x % NB, b = blockIdx.x / NB;
    const fbbev_msda_band bd = fbbev_msda_band_of(band, L, ss, lsi, budget);
    int lo = 0x7fffffff, hi = -1;
    const unsigned int* rr = ranges + ((long long)b * L + bd.level) * Q;
    constexpr int U = 8;                                   // (round 6: eight range words in flight per thread; one per iteration was 156 round trips at Q = 40 000)
    for (int q0 = threadIdx.x; q0 < Q; q0 += 256 * U) {
        unsigned int vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = rr[q0 + 256 * u < Q ? q0 + 256 * u : q0];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + 256 * u;
            const int first = (int)(vv[u] & 0xffffu), last = (int)(vv[u] >> 16);
            if (q < Q && first <= last && first < bd.r1 && last >= bd.r0) { lo = q < lo ? q : lo; hi = q > hi ? q : hi; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = red[0][w] < lo ? red[0][w] : lo; hi = red[1][w] > hi ? red[1][w] : hi; }
        qrange[((long long)b * NB + band) * 2] = hi >= 0 ? lo : 0;
        qrange[((long long)b * NB + band) * 2 + 1] = hi >= 0 ? hi + 1 : 0;
    }
}

template <int NT, int DH>
__global__ void __launch_bounds__(NT)
k_msda_bwd_scatter(const int64_t* __restrict__ ss, const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                   const float* __restrict__ attn, const float* __restrict__ grad_out, const unsigned int* __restrict__ ranges,
                   const int* __restrict__ qrange, int S, int M, int L, int Q, int P, int NB, int budget,
                   float* __restrict__ grad_value) {
    long long* plane = reinterpret_cast<long long*>(fbbev_dyn_lds_f32());
    __shared__ float red[2][NT / 64];
    const int band = blockIdx.x % NB;
    const int m = (blockIdx.x / NB) % M;
    const int b = blockIdx.x / (NB * M);
    const fbbev_msda_band bd = fbbev_msda_band_of(band, L, ss, lsi, budget);
    const int ntok = (bd.r1 - bd.r0) * bd.w, tok0 = bd.r0 * bd.w;
    const int plane_w = FBBEV_DA_PLANE_WORDS(ntok, DH);
    const int qlo = qrange[((long long)b * NB + band) * 2], qhi = qrange[((long long)b * NB + band) * 2 + 1];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < plane_w; i += NT) plane[i] = 0ll;
    // scale of the plane from the largest |grad_output| and |attention weight| of the span
    float gmax = 0.f, amax = 1.f;
    bool finite = true;
    // (round 6: one thread per query of the span, its DH gradient floats and P weights requested together -- as two element-wise loops
    // of one load per iteration the scan was ~90 dependent round trips per workgroup, most of the kernel's time)
    for (int q0 = qlo; q0 < qhi; q0 += NT) {
        const int q_ = q0 + (int)threadIdx.x, q = q_ < qhi ? q_ : qhi - 1;
        const long long u = ((long long)b * Q + q) * M + m;
        float gv[DH], av[4];
#pragma unroll
        for (int c = 0; c < DH; ++c) gv[c] = grad_out[u * DH + c];
        const long long wp = (u * L + bd.level) * P;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = attn[wp + (e < P ? e : 0)];
        float am = 0.f;
        for (int e = 4; e < P; ++e) am = fmaxf(am, fabsf(attn[wp + e]));                 // (P > 4: not an FB-OCC shape)
#pragma unroll
        for (int c = 0; c < DH; ++c) { const float v = fabsf(gv[c]); finite = finite && (v < __builtin_inff()); gmax = fmaxf(gmax, v); }
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(av[e]));
        finite = finite && (am < __builtin_inff());
        amax = fmaxf(amax, am);
    }
    if (!finite) gmax = __builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64)); amax = fmaxf(amax, __shfl_xor(amax, o, 64)); }
    if (lane == 0) { red[0][threadIdx.x >> 6] = gmax; red[1][threadIdx.x >> 6] = amax; }
    __syncthreads();
    gmax = red[0][0]; amax = red[1][0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) { gmax = fmaxf(gmax, red[0][w]); amax = fmaxf(amax, red[1][w]); }
    gmax = gmax * amax;
    const bool poisoned = !(gmax < __builtin_inff());
    float sc = 0.f, inv_sc = 0.f;
    if (!poisoned && gmax > 0.f) {
        unsigned int gb;
        __builtin_memcpy(&gb, &gmax, 4);
        int ex = (int)((gb >> 23) & 255u) - 126;
        if (ex < -90) ex = -90;
        if (ex > 96) ex = 96;
        const unsigned int sb = (unsigned int)(127 + 30 - ex) << 23, ib = (unsigned int)(127 - 30 + ex) << 23;
        __builtin_memcpy(&sc, &sb, 4);
        __builtin_memcpy(&inv_sc, &ib, 4);
    }
    const unsigned int* rr = ranges + ((long long)b * L + bd.level) * Q;
    if (!poisoned && sc > 0.f) {
        // round 6: a query's words -- row range, upstream gradient, the level's locations and weights -- are requested together
        // (clamped, unconditional) before any is used: as written first (range -> gradient -> per point location / weight) every query
        // cost 2 + P dependent round trips in front of its LDS adds
        constexpr int PB = 4;
        for (int q0 = qlo; q0 < qhi; q0 += NT) {
            const int q_ = q0 + (int)threadIdx.x, q = q_ < qhi ? q_ : qhi - 1;
            const unsigned int v = rr[q];
            const long long u = ((long long)b * Q + q) * M + m;
            const long long wp = (u * L + bd.level) * P;
            float gs[DH];
#pragma unroll
            for (int c = 0; c < DH; ++c) gs[c] = grad_out[u * DH + c];
            float lw[PB], lh[PB], wt[PB];
#pragma unroll
            for (int e = 0; e < PB; ++e) {
                const int pp = e < P ? e : 0;
                lw[e] = loc[(wp + pp) * 2]; lh[e] = loc[(wp + pp) * 2 + 1]; wt[e] = attn[wp + pp];
            }
            const int first = (int)(v & 0xffffu), last = (int)(v >> 16);
            if (q_ >= qhi || !(first <= last && first < bd.r1 && last >= bd.r0)) continue;
#pragma unroll
            for (int c = 0; c < DH; ++c) gs[c] = gs[c] * sc;                              // sc is a power of two: exact
            for (int p = 0; p < P; ++p) {
                float loc_w, loc_h, weight;
                if (p < PB) {                                                               // (compile-time after unrolling for P <= 4)
                    loc_w = p == 0 ? lw[0] : p == 1 ? lw[1] : p == 2 ? lw[2] : lw[3];
                    loc_h = p == 0 ? lh[0] : p == 1 ? lh[1] : p == 2 ? lh[2] : lh[3];
                    weight = p == 0 ? wt[0] : p == 1 ? wt[1] : p == 2 ? wt[2] : wt[3];
                } else {
                    loc_w = loc[(wp + p) * 2]; loc_h = loc[(wp + p) * 2 + 1]; weight = attn[wp + p];
                }
                const float h_im = loc_h * bd.h - 0.5f, w_im = loc_w * bd.w - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)bd.h && w_im < (float)bd.w)) continue;
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, bd.h, bd.w, 1);       // o1..o4 = token index in the level
                const int t1 = s.o1 - tok0, t2 = s.o2 - tok0, t3 = s.o3 - tok0, t4 = s.o4 - tok0;
                const bool k1 = s.o1 >= 0 && t1 >= 0 && t1 < ntok, k2 = s.o2 >= 0 && t2 >= 0 && t2 < ntok;
                const bool k3 = s.o3 >= 0 && t3 >= 0 && t3 < ntok, k4 = s.o4 >= 0 && t4 >= 0 && t4 < ntok;
                float tg[DH];                                                         // the reference's top_grad * attn_weight
#pragma unroll
                for (int c = 0; c < DH; ++c) tg[c] = gs[c] * weight;
                // corner outer, channel inner: one divergent region per corner (see k_da_cross_attn_bwd_scatter)
                fbbev_lds_corner_add<DH>(k1, plane + FBBEV_DA_PLANE_IDX(t1, DH), s.w1, tg);
                fbbev_lds_corner_add<DH>(k2, plane + FBBEV_DA_PLANE_IDX(t2, DH), s.w2, tg);
                fbbev_lds_corner_add<DH>(k3, plane + FBBEV_DA_PLANE_IDX(t3, DH), s.w3, tg);
                fbbev_lds_corner_add<DH>(k4, plane + FBBEV_DA_PLANE_IDX(t4, DH), s.w4, tg);
            }
        }
    }
    __syncthreads();
    float* dst = grad_value + (((long long)b * S + bd.ls + tok0) * M + m) * DH;
    for (int i = threadIdx.x; i < ntok * DH; i += NT) {
        const int t = i / DH, c = i - t * DH;
        const long long acc = plane[FBBEV_DA_PLANE_IDX(t, DH) + c];
        dst[(long long)t * M * DH + c] = poisoned ? __builtin_nanf("") : (float)acc * inv_sc;      // one rounding (int64 -> fp32)
    }
}

// Unit-owned gradients with ONE LANE PER (b, q, head) UNIT (the forward's unit-per-lane shape): the four corners of a sample
// are read as DH/2 8-byte loads each (a head's DH floats are 8-byte aligned for even DH) and the three sums stay in the lane --
// no 16-lane groups with 6 idle lanes at DH = 10, no shuffles.  Same per-channel expressions as k_msda_bwd, channels summed
// in ascending order (k_msda_bwd sums 16-lane partials: the results agree to fp32 rounding).
template <int DH>
__global__ void __launch_bounds__(256)
k_msda_bwd_unit(long long n_units, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                const int64_t* __restrict__ level_start, const float* __restrict__ loc, const float* __restrict__ attn,
                const float* __restrict__ grad_out, int spatial_size, int M, int L, int Q, int P,
                float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
    static_assert(DH % 2 == 0, "8-byte corner loads");
    const long long unit = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (unit >= n_units) return;
    const int m = (int)(unit % M);
    const long long b = unit / M / Q;
    const int row_stride = M * DH;
    float top[DH];
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        const fbbev_v2f t = *reinterpret_cast<const fbbev_v2f*>(grad_out + unit * DH + c);
        top[c] = t[0]; top[c + 1] = t[1];
    }
    // round 6: the unit's words of up to four points (location, weight, the gradients accumulated so far) are requested together, then
    // the corner runs of two samples at a time -- as one point per iteration (location -> corner addresses -> corners -> read-add-write
    // of the gradients) the loop was three dependent round trips per point (0.29 ms for the BEV self-attention of BASELINE configs[2]);
    // the same expressions in the same order: identical bits
    constexpr int PB = 4;
    const long long wp0 = unit * L * P;
    for (int l = 0; l < L; ++l) {
        const int height = (int)spatial_shapes[2 * l], width = (int)spatial_shapes[2 * l + 1];
        const float* vb = value + (b * spatial_size + level_start[l]) * row_stride + m * DH;
        for (int p0 = 0; p0 < P; p0 += PB) {
            fbbev_v2f lc[PB], gl[PB];
            float wt[PB], ga[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const long long wp = wp0 + (long long)l * P + (p0 + u < P ? p0 + u : p0);          // (clamped: unconditional loads)
                lc[u] = *reinterpret_cast<const fbbev_v2f*>(loc + wp * 2);
                gl[u] = *reinterpret_cast<const fbbev_v2f*>(grad_loc + wp * 2);
                wt[u] = attn[wp];
                ga[u] = grad_attn[wp];
            }
            float rw[PB], rx[PB], ry[PB];
#pragma unroll
            for (int u0 = 0; u0 < PB; u0 += 2) {
                fbbev_bilinear s[2];
                bool live[2], k1[2], k2[2], k3[2], k4[2];
                fbbev_v2f a1[2][DH / 2], a2[2][DH / 2], a3[2][DH / 2], a4[2][DH / 2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = u0 + e;
                    const float h_im = lc[u][1] * height - 0.5f, w_im = lc[u][0] * width - 0.5f;
                    live[e] = p0 + u < P && h_im > -1.f && w_im > -1.f && h_im < (float)height && w_im < (float)width;
                    s[e] = fbbev_bilinear_setup(live[e] ? h_im : 0.f, live[e] ? w_im : 0.f, height, width, row_stride);
                    k1[e] = live[e] && s[e].o1 >= 0; k2[e] = live[e] && s[e].o2 >= 0; k3[e] = live[e] && s[e].o3 >= 0; k4[e] = live[e] && s[e].o4 >= 0;
                    const float* p1 = vb + (k1[e] ? s[e].o1 : 0);
                    const float* p2 = vb + (k2[e] ? s[e].o2 : 0);
                    const float* p3 = vb + (k3[e] ? s[e].o3 : 0);
                    const float* p4 = vb + (k4[e] ? s[e].o4 : 0);
#pragma unroll
                    for (int k = 0; k < DH / 2; ++k) {            // unconditional loads (a padded corner reads token 0), zeros selected below
                        a1[e][k] = *reinterpret_cast<const fbbev_v2f*>(p1 + 2 * k);
                        a2[e][k] = *reinterpret_cast<const fbbev_v2f*>(p2 + 2 * k);
                        a3[e][k] = *reinterpret_cast<const fbbev_v2f*>(p3 + 2 * k);
                        a4[e][k] = *reinterpret_cast<const fbbev_v2f*>(p4 + 2 * k);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = u0 + e;
                    const float weight = wt[u];
                    float g_w = 0.f, g_x = 0.f, g_y = 0.f;
#pragma unroll
                    for (int c = 0; c < DH; ++c) {
                        const float v1 = k1[e] ? a1[e][c >> 1][c & 1] : 0.f, v2 = k2[e] ? a2[e][c >> 1][c & 1] : 0.f;
                        const float v3 = k3[e] ? a3[e][c >> 1][c & 1] : 0.f, v4 = k4[e] ? a4[e][c >> 1][c & 1] : 0.f;
                        const float tgv = top[c] * weight;
                        const float ghw = -s[e].hw * v1 - s[e].lw * v2 + s[e].hw * v3 + s[e].lw * v4;
                        const float gww = -s[e].hh * v1 + s[e].hh * v2 - s[e].lh * v3 + s[e].lh * v4;
                        g_w += top[c] * (s[e].w1 * v1 + s[e].w2 * v2 + s[e].w3 * v3 + s[e].w4 * v4);
                        g_x += (float)width * gww * tgv;
                        g_y += (float)height * ghw * tgv;
                    }
                    rw[u] = g_w; rx[u] = g_x; ry[u] = g_y;
                    if (!live[e]) { rw[u] = 0.f; rx[u] = 0.f; ry[u] = 0.f; }
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (p0 + u >= P) break;
                const long long wp = wp0 + (long long)l * P + p0 + u;
                // (a sample outside the image contributed nothing before either: the words keep their values)
                grad_attn[wp] = ga[u] + rw[u];
                *reinterpret_cast<fbbev_v2f*>(grad_loc + wp * 2) = fbbev_v2f{gl[u][0] + rx[u], gl[u][1] + ry[u]};
            }
        }
    }
}
