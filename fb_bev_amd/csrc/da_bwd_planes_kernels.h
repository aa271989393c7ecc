// Round 4: the unit-owned gradients of the DA cross-attention backward (sampling offsets, attention weights, depth distribution)
// on HEAD PLANES with the forward's mapping (da_fused_kernels.h): camera tokens as (B*Ncam, M, S, DH) planes, a workgroup = the
// MH heads of a patch of 64 BEV queries, a wave = one head, a lane = one query.  k_da_cross_attn_bwd_unit (da_kernels.h) gives a
// lane one (query, head) unit of a row-major token buffer: 64 lanes of a load touch 64 scattered rows of 8 heads, and it was the
// largest kernel of the DA backward (1.3 ms at the configs[2] pyramid).  Here the two x-corners of a sample are one run of
// 2*DH floats of ONE head plane, neighbouring lanes read neighbouring runs, and the per-(camera, query) work -- hit test,
// reference points, depth weights -- is done once per workgroup into LDS instead of once per head.
//   d slots / d attention = depth_weight * <g, sample>;  d slots / d offset = attention * depth_weight * <g, d sample / d (x, y)>;
//   d slots / d depth_weight = attention * <g, sample>, pushed to the four taps of the query's depth-bin plane by fp32 atomics
//   after a fixed-order sum over the workgroup's heads (as k_da_cross_attn_bwd_unit: spatial_cross_attention_depth.py:136-223,
//   513-595 differentiated; mmcv ms_deform_attn_backward's col2im arithmetic for the sample).
// Loop order: level outer, camera inner (uniform over the WORKGROUP: a hit depends on (query, camera) only, so every wave walks the
// same cameras and the level loop may hold barriers).  The per-unit tensors are touched by the workgroup TOGETHER: a lane's
// offsets / attention / gradient words lie 2 KB from its neighbour's (lane = query), but the 64 queries x 8 heads x 8 points of a
// level are whole contiguous runs -- they are staged through two LDS tiles with coalesced loads, the level's gradients go back
// through the same tiles and are stored to grad_offsets / grad_attn with coalesced writes (the first form of this
// kernel let every lane touch its own words: 1.72 ms against the row kernel's 1.31).  Cameras add in ascending order.
// A padded corner keeps a valid address and gets weight AND slope 0; a sample outside the image contributes nothing.
#pragma once
#include "rt.h"
#include "da_kernels.h"
#include "da_fused_kernels.h"

template <int DH>
struct fbbev_dbp_pending {
    static constexpr int NV = (2 * DH) / 4;
    fbbev_v4f a[NV], b[NV];            // the two row runs: tokens (x, x+1) of rows y0 and y1
    float sx0, sx1, sy0, sy1;          // bilinear weights of the run's two tokens / of the two rows
    int edge;                          // bit 0 left, 1 right, 2 top, 3 bottom, 4 live: the slopes of the weights in x / y follow from
                                       // it (-1, +1 inside; a padded corner has slope 0) -- four registers less per slot than the floats
};

template <int DH>
__device__ __forceinline__ void fbbev_dbp_issue(const char* __restrict__ plane, int level_off /* floats */, float h_im, float w_im,
                                                int sh, int sw, bool enable, fbbev_dbp_pending<DH>& p) {
    const bool live = enable && h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw;
    const float h = live ? h_im : 0.f, w = live ? w_im : 0.f;
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const float lh = h - (float)h_low, lw = w - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const float on = live ? 1.f : 0.f;
    // x: the run covers tokens (xb, xb + 1), clamped into the row (fbbev_daf_issue): at the left edge the only valid corner (the
    // HIGH one, weight lw, slope +1) sits in slot 0, at the right edge the LOW one (weight hw, slope -1) in slot 1
    const bool left = w_low < 0, right = w_low >= sw - 1;
    const int xb = left ? 0 : (right ? sw - 2 : w_low);
    p.sx0 = on * (left ? lw : (right ? 0.f : hw)); p.sx1 = on * (left ? 0.f : (right ? hw : lw));
    const bool top = h_low < 0, bottom = h_low >= sh - 1;
    const int y0 = top ? 0 : h_low, y1 = bottom ? h_low : h_low + 1;
    p.sy0 = on * (top ? 0.f : hh); p.sy1 = on * (bottom ? 0.f : lh);
    p.edge = (left ? 1 : 0) | (right ? 2 : 0) | (top ? 4 : 0) | (bottom ? 8 : 0) | (live ? 16 : 0);
    const unsigned o0 = (unsigned)(level_off + (y0 * sw + xb) * DH) * 4u, o1 = (unsigned)(level_off + (y1 * sw + xb) * DH) * 4u;
    constexpr int NV = (2 * DH) / 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        fbbev_v4f t0, t1;
        __builtin_memcpy(&t0, plane + o0 + 16 * k, 16);
        __builtin_memcpy(&t1, plane + o1 + 16 * k, 16);
        p.a[k] = t0; p.b[k] = t1;
    }
}

// <g, sample>, <g, d sample / dx>, <g, d sample / dy> of one pending sample
template <int DH>
__device__ __forceinline__ void fbbev_dbp_consume(const fbbev_dbp_pending<DH>& p, const fbbev_v2f (&g)[DH / 2], float& dot, float& gx,
                                                  float& gy) {
    fbbev_v2f d2 = {0.f, 0.f}, x2 = {0.f, 0.f}, y2 = {0.f, 0.f};
    const bool live = (p.edge & 16) != 0, left = (p.edge & 1) != 0, right = (p.edge & 2) != 0, top = (p.edge & 4) != 0, bottom = (p.edge & 8) != 0;
    const float on = live ? 1.f : 0.f;
    const float dx0 = on * (left ? 1.f : (right ? 0.f : -1.f)), dx1 = on * (left ? 0.f : (right ? -1.f : 1.f));
    const float dy0 = on * (top ? 0.f : -1.f), dy1 = on * (bottom ? 0.f : 1.f);
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        const fbbev_v2f v00 = fbbev_daf_pair<DH>(p.a, 0, c), v01 = fbbev_daf_pair<DH>(p.a, 1, c);
        const fbbev_v2f v10 = fbbev_daf_pair<DH>(p.b, 0, c), v11 = fbbev_daf_pair<DH>(p.b, 1, c);
        const fbbev_v2f r0 = p.sx0 * v00 + p.sx1 * v01, r1 = p.sx0 * v10 + p.sx1 * v11;       // the two rows blended in x
        const fbbev_v2f t0 = dx0 * v00 + dx1 * v01, t1 = dx0 * v10 + dx1 * v11;       // ... and their slopes in x
        d2 += g[c / 2] * (p.sy0 * r0 + p.sy1 * r1);
        x2 += g[c / 2] * (p.sy0 * t0 + p.sy1 * t1);
        y2 += g[c / 2] * (dy0 * r0 + dy1 * r1);
    }
    dot = d2[0] + d2[1]; gx = x2[0] + x2[1]; gy = y2[0] + y2[1];
}

// value rows (B*Ncam, S, M, HS) -- plain (head m at m*HS) or chunk-major interleaved ((HS/4, M, 4), head_minor bit 2) -> head planes
__global__ void __launch_bounds__(256)
k_value_rows_to_head_planes(const float* __restrict__ rows, long long n_tok, int S, int M, int DH, int HS, int interleaved,
                            float* __restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = n_tok * M * DH;
    if (i >= n) return;
    const int c = (int)(i % DH);
    const long long t = i / DH;
    const int m = (int)(t % M);
    const long long r = t / M;                                                     // token row
    const long long bn = r / S, tok = r - bn * S;
    const long long src = r * (long long)(M * HS) + (interleaved ? (c >> 2) * (M * 4) + m * 4 + (c & 3) : m * HS + c);
    planes[((bn * M + m) * S + tok) * DH + c] = rows[src];
}

#define FBBEV_DBP_QC 13          // floats of a (camera, query) record: rx[4] ry[4] dw[4] hit
#define FBBEV_DBP_OQ (FBBEV_DAF_P * 8 * 2 + 2)   // floats per query of the offsets staging tile [q][p][m][xy] (+2: lanes 8 bytes apart in banks)
#define FBBEV_DBP_AQ (8 * FBBEV_DAF_P + 1)       // floats per query of the attention staging tile [q][m][p] (+1)
// LDS: (camera, query) records | offsets tile | attention tile | attention-gradient tile | d/d depth-weight slots [head][camera][query][anchor]
__host__ __device__ inline size_t fbbev_dbp_lds_bytes(int MH, int Ncam) {
    return (size_t)Ncam * 64 * FBBEV_DBP_QC * 4 + (size_t)64 * FBBEV_DBP_OQ * 4 + (size_t)2 * 64 * FBBEV_DBP_AQ * 4 +
           (size_t)MH * Ncam * 64 * FBBEV_DAF_ZA * 4;
}

// planes (B*Ncam, MH, S, DH); offsets / attn / grad_offsets / grad_attn in the layouts of k_da_cross_attn_bwd_unit (head_minor bits
// 0 / 1); grad_slots (B, Q, MH*DH); patch = 8 x 8 queries of a bev_w-wide grid (bev_w > 0) or 64 consecutive queries (bev_w == 0).
template <int DH, int MH>
__global__ void __launch_bounds__(64 * MH)
k_da_bwd_unit_planes(const float* __restrict__ planes, const int64_t* __restrict__ spatial_shapes,
                     const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                     const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                     const float* __restrict__ qdepth, const float* __restrict__ offsets, const float* __restrict__ attn,
                     const float* __restrict__ grad_slots, int B, int Ncam, int S, int L, int Q, int bev_w, int DC, float d0,
                     float dstep, int head_minor, float* __restrict__ grad_pred_depth, float* __restrict__ grad_offsets,
                     float* __restrict__ grad_attn, unsigned int* __restrict__ gmax_bits) {
    constexpr int P = FBBEV_DAF_P, ZA = FBBEV_DAF_ZA, NT = 64 * MH;
    static_assert((2 * DH) % 4 == 0 && DH % 2 == 0, "runs of whole 16-byte pieces, channel pairs");
    static_assert(MH == 8, "the staging tiles hold 8 heads");
    float* qc = fbbev_dyn_lds_f32();                                                        // [Ncam][64][QC]
    float* off_s = qc + (size_t)Ncam * 64 * FBBEV_DBP_QC;                                   // [64][OQ]: (p, m, xy) of a query
    float* att_s = off_s + 64 * FBBEV_DBP_OQ;                                               // [64][AQ]: (m, p) of a query
    float* gat_s = att_s + 64 * FBBEV_DBP_AQ;                                               // [64][AQ]: the level's d/d attention
    float* dd = gat_s + 64 * FBBEV_DBP_AQ;                                                  // [MH][Ncam][64][ZA]
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    // patches: 8 x 8 on a grid, 64 x 1 on a list
    const int gw = bev_w > 0 ? bev_w : Q, gh = Q / gw, plog = bev_w > 0 ? 3 : 6, pw = 1 << plog, ph = 64 >> plog;
    const int pxn = (gw + pw - 1) / pw, pyn = (gh + ph - 1) / ph;
    const long long n_wg = (long long)B * pxn * pyn, per_xcd = (n_wg + 7) / 8;
    const long long wgid = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);       // XCD-contiguous patch order
    if (wgid >= n_wg) return;                                                               // uniform
    const int b = (int)(wgid / ((long long)pxn * pyn)), pi = (int)(wgid - (long long)b * pxn * pyn);
    const int py = pi / pxn, px = pi - py * pxn;
    const int x0 = px * pw, y0 = py * ph;
    // ---------------- phase A (whole workgroup): the patch's (camera, query) records (k_da_cross_attn_fused's)
    for (int i = threadIdx.x; i < Ncam * 64; i += NT) {
        const int cam = i >> 6, ql = i & 63;
        const int qy = y0 + (ql >> plog), qx = x0 + (ql & (pw - 1));
        const bool inb = qy < gh && qx < gw;
        float* rec = qc + (size_t)i * FBBEV_DBP_QC;
        const long long base = (((long long)cam * B + b) * Q + (inb ? (long long)qy * gw + qx : 0)) * ZA;
        unsigned int mask4;
        __builtin_memcpy(&mask4, mask + base, 4);                                       // ZA = 4 mask bytes
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
        rec[3 * ZA] = (inb && mask4 != 0u) ? 1.f : 0.f;
    }
    __syncthreads();
    // ---------------- phase B (wave = head m, lane = query)
    const int m = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int qy = y0 + (lane >> plog), qx = x0 + (lane & (pw - 1));
    const bool valid = qy < gh && qx < gw;
    const long long bq = (long long)b * Q + (valid ? (long long)qy * gw + qx : 0);
    const long long u = bq * MH + m;
    const float* my_qc = qc + (size_t)lane * FBBEV_DBP_QC;
    int count = 0;
    for (int cam = 0; cam < Ncam; ++cam) count += (fbbev_lds_ld_f32(my_qc + (size_t)cam * 64 * FBBEV_DBP_QC + 3 * ZA) != 0.f) ? 1 : 0;
    fbbev_v2f g[DH / 2];
    {
        float gm = 0.f;
        bool fin = true;
        const float inv = (float)(count > 1 ? count : 1);
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) {
            fbbev_v2f t = {0.f, 0.f};
            if (valid) t = *reinterpret_cast<const fbbev_v2f*>(grad_slots + u * DH + 2 * c);
            const float a0 = fabsf(t[0]), a1 = fabsf(t[1]);
            fin = fin && (a0 < __builtin_inff()) && (a1 < __builtin_inff());
            gm = fmaxf(gm, fmaxf(a0, a1));
            g[c][0] = t[0] / inv; g[c][1] = t[1] / inv;
        }
        if (gmax_bits) {                                                                // the sample's max |gradient| (see k_da_bwd_scatter_owned)
            if (!fin) gm = __builtin_inff();
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
            unsigned int gb;
            __builtin_memcpy(&gb, &gm, 4);
            if (lane == 0 && gb != 0u) atomicMax(gmax_bits + b, gb);                       // per SAMPLE (round 5): a workgroup = one sample's patch
        }
    }
    const int LP = L * P;
    const char* pb = reinterpret_cast<const char*>(planes);
    for (int i = threadIdx.x; i < MH * Ncam * 64 * ZA; i += NT) dd[i] = 0.f;
    // cooperative pass over a level's unit words: item i = (query ql, point p, head h) in the order of the head-minor layouts.
    // Round 6: ONE pass per level boundary -- the next level's words are REQUESTED first (all of a thread's items, clamped addresses),
    // the finished level's gradients go back from the tiles while they are in flight, then the loaded words take the gradients'
    // slots (an item's slots are touched by this thread only: no barrier in between).  As three loops of load / wait / LDS write per
    // item the staging was 8 dependent round trips per level and two more barriers (tools/isa_chains.py: (47, 2, 2), (71, 2, 2)).
    constexpr int NI = (64 * P * MH + NT - 1) / NT;
    auto unit_words = [&](int l_back, int l_load) {          // -1: nothing to write back / to load
        fbbev_v2f ot[NI];
        float at[NI];
        long long io_b[NI], ia_b[NI];
        bool ok[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = (int)threadIdx.x + k * NT;
            const int ql = i / (P * MH), r = i - ql * (P * MH), p = r / MH, h = r - p * MH;
            const int uy = y0 + (ql >> plog), ux = x0 + (ql & (pw - 1));
            ok[k] = i < 64 * P * MH && uy < gh && ux < gw;
            const long long ubq = (long long)b * Q + (ok[k] ? (long long)uy * gw + ux : 0), uu = ubq * MH + h;
            const long long rowo = (head_minor & 1) ? ubq * LP * MH + h : uu * LP, rowa = (head_minor & 2) ? ubq * LP * MH + h : uu * LP;
            const long long so_ = (head_minor & 1) ? MH : 1, sa_ = (head_minor & 2) ? MH : 1;
            io_b[k] = (rowo + (long long)((l_back < 0 ? 0 : l_back) * P + p) * so_) * 2;
            ia_b[k] = rowa + (long long)((l_back < 0 ? 0 : l_back) * P + p) * sa_;
            if (l_load >= 0) {
                const long long io = (rowo + (long long)(l_load * P + p) * so_) * 2, ia = rowa + (long long)(l_load * P + p) * sa_;
                ot[k] = *reinterpret_cast<const fbbev_v2f*>(offsets + io);
                at[k] = attn[ia];
            }
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = (int)threadIdx.x + k * NT;
            const int ql = i / (P * MH), r = i - ql * (P * MH), p = r / MH, h = r - p * MH;
            float* so = off_s + ql * FBBEV_DBP_OQ + (p * MH + h) * 2;
            float* sa = att_s + ql * FBBEV_DBP_AQ + h * P + p;
            if (!ok[k]) continue;
            if (l_back >= 0) {
                // the level's gradients, STORED: the entry's contract is "accumulated into caller-zeroed tensors", every word of a
                // valid query has exactly this one writer per call (a unit no camera sees stores its zeros), so a store is the
                // accumulation -- without reading 492 MB of zeros back at the configs[2] pyramid
                *reinterpret_cast<fbbev_v2f*>(grad_offsets + io_b[k]) = fbbev_v2f{so[0], so[1]};
                grad_attn[ia_b[k]] = gat_s[ql * FBBEV_DBP_AQ + h * P + p];
            }
            if (l_load >= 0) { so[0] = ot[k][0]; so[1] = ot[k][1]; sa[0] = at[k]; }
        }
    };
    float* my_off = off_s + lane * FBBEV_DBP_OQ + m * 2;                // + p * MH * 2
    float* my_att = att_s + lane * FBBEV_DBP_AQ + m * P;               // + p
    float* my_gat = gat_s + lane * FBBEV_DBP_AQ + m * P;
    float* my_dd = dd + (((size_t)m * Ncam) * 64 + lane) * ZA;          // + cam * 64 * ZA + z
    unit_words(-1, 0);
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
        const float fsh = (float)sh, fsw = (float)sw;
        const int lvl_off = (int)level_start[l] * DH;
        // the lane's offsets / weights of the level move to registers; their slots become the level's gradient accumulators
        // (plain read-add-write by the owner lane -- registers are needed for the two samples in flight: 80 of them)
        fbbev_v2f o[P];
#pragma unroll
        for (int p = 0; p < P; ++p) { o[p][0] = fbbev_lds_ld_f32(my_off + p * MH * 2); o[p][1] = fbbev_lds_ld_f32(my_off + p * MH * 2 + 1); }
#pragma unroll
        for (int p = 0; p < P; ++p) { my_off[p * MH * 2] = 0.f; my_off[p * MH * 2 + 1] = 0.f; my_gat[p] = 0.f; }
        for (int cam = 0; cam < Ncam; ++cam) {
            const float* rec = my_qc + (size_t)cam * 64 * FBBEV_DBP_QC;
            const bool hit = valid && fbbev_lds_ld_f32(rec + 3 * ZA) != 0.f;
            if (__ballot(hit) == 0ull) continue;                       // the same for every wave of the workgroup
            float rx[ZA], ry[ZA], dw[ZA];
#pragma unroll
            for (int z = 0; z < ZA; ++z) {
                rx[z] = fbbev_lds_ld_f32(rec + z); ry[z] = fbbev_lds_ld_f32(rec + ZA + z); dw[z] = fbbev_lds_ld_f32(rec + 2 * ZA + z);
            }
            float* slot = my_dd + (size_t)cam * 64 * ZA;                // the lane's own words: no other lane touches them
            const char* plane = pb + (((long long)b * Ncam + cam) * MH + m) * (long long)S * DH * 4;      // wave-uniform base
            fbbev_dbp_pending<DH> pend[2];
            auto start = [&](int p, fbbev_dbp_pending<DH>& slot) {
                const int z = p % ZA;
                const float loc_w = rx[z] + __fdiv_rn(o[p][0], fsw), loc_h = ry[z] + __fdiv_rn(o[p][1], fsh);
                fbbev_dbp_issue<DH>(plane, lvl_off, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, sh, sw, hit, slot);
            };
            start(0, pend[0]);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (p + 1 < P) start(p + 1, pend[(p + 1) & 1]);
                fbbev_sched_fence();
                float dot, gx, gy;
                fbbev_dbp_consume<DH>(pend[p & 1], g, dot, gx, gy);     // all zero for a lane the camera does not see
                fbbev_sched_fence();
                const int z = p % ZA;
                const float av = fbbev_lds_ld_f32(my_att + p), weight = av * dw[z];
                my_gat[p] += dw[z] * dot;
                my_off[p * MH * 2] += weight * gx; my_off[p * MH * 2 + 1] += weight * gy;
                slot[z] += av * dot;                                    // 0 for a lane the camera does not see
            }
        }
        __syncthreads();                                               // the tiles now hold the level's gradients
        unit_words(l, l + 1 < L ? l + 1 : -1);
        if (l + 1 < L) __syncthreads();
    }
    __syncthreads();
    // d / d depth weight: summed over the heads in head order, wave z pushes anchor z of the 64 queries to the four taps of the
    // query's bin plane (fp32 atomics, as k_da_cross_attn_bwd_unit after its head reduction)
    if (m < ZA && valid) {
        const int z = m;
        for (int cam = 0; cam < Ncam; ++cam) {
            const float* rec = my_qc + (size_t)cam * 64 * FBBEV_DBP_QC;
            if (fbbev_lds_ld_f32(rec + 3 * ZA) == 0.f) continue;
            float dsum = 0.f;
#pragma unroll
            for (int h = 0; h < MH; ++h) dsum += dd[((((size_t)h * Ncam) + cam) * 64 + lane) * ZA + z];
            if (dsum == 0.f) continue;
            const long long bn = (long long)b * Ncam + cam;
            const long long base = (((long long)cam * B + b) * Q + ((long long)qy * gw + qx)) * ZA;
            float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
            fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
            int off[4];
            float wgt[4];
            fbbev_daf_plane_corners(fbbev_lds_ld_f32(rec + z), fbbev_lds_ld_f32(rec + ZA + z), H0, W0, off, wgt);   // z is the wave's: no register indexing
            float* gd = grad_pred_depth + (bn * DC + (int)fb) * (long long)(H0 * W0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (wgt[k] != 0.f) fbbev_atomic_add_f32(gd + off[k], wgt[k] * dsum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t fbbev_dfp_lds_bytes(int Ncam) {
    return (size_t)Ncam * 64 * FBBEV_DBP_QC * 4 + (size_t)64 * FBBEV_DBP_OQ * 4 + (size_t)64 * FBBEV_DBP_AQ * 4;
}

// The TRAINING forward on head planes: projected offsets / softmaxed weights come from memory (autograd owns the projections),
// everything else is k_da_cross_attn_fused's sampler -- records once per workgroup, wave = head, lane = query of the patch, two
// samples in flight -- with the level's unit words staged through the LDS tiles of k_da_bwd_unit_planes.  Replaces
// k_da_cross_attn_fwd_unit (0.71 ms at the configs[2] pyramid) when autograd is on.  slots (B, Q, MH*DH), written.
template <int DH, int MH>
__global__ void __launch_bounds__(64 * MH)
k_da_fwd_planes(const float* __restrict__ planes, const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
                const float* __restrict__ pred_depth, const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                const float* __restrict__ qdepth, const float* __restrict__ offsets, const float* __restrict__ attn, int B, int Ncam,
                int S, int L, int Q, int bev_w, int DC, float d0, float dstep, int head_minor, float* __restrict__ slots) {
    constexpr int P = FBBEV_DAF_P, ZA = FBBEV_DAF_ZA, NT = 64 * MH;
    static_assert(MH == 8, "the staging tiles hold 8 heads");
    float* qc = fbbev_dyn_lds_f32();                                                        // [Ncam][64][QC]
    float* off_s = qc + (size_t)Ncam * 64 * FBBEV_DBP_QC;                                   // [64][OQ]: (p, m, xy) of a query
    float* att_s = off_s + 64 * FBBEV_DBP_OQ;                                               // [64][AQ]: (m, p) of a query
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int gw = bev_w > 0 ? bev_w : Q, gh = Q / gw, plog = bev_w > 0 ? 3 : 6, pw = 1 << plog, ph = 64 >> plog;
    const int pxn = (gw + pw - 1) / pw, pyn = (gh + ph - 1) / ph;
    const long long n_wg = (long long)B * pxn * pyn, per_xcd = (n_wg + 7) / 8;
    const long long wgid = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);       // XCD-contiguous patch order
    if (wgid >= n_wg) return;                                                               // uniform
    const int b = (int)(wgid / ((long long)pxn * pyn)), pi = (int)(wgid - (long long)b * pxn * pyn);
    const int py = pi / pxn, px = pi - py * pxn;
    const int x0 = px * pw, y0 = py * ph;
    for (int i = threadIdx.x; i < Ncam * 64; i += NT) {                                     // the records of k_da_cross_attn_fused
        const int cam = i >> 6, ql = i & 63;
        const int qy = y0 + (ql >> plog), qx = x0 + (ql & (pw - 1));
        const bool inb = qy < gh && qx < gw;
        float* rec = qc + (size_t)i * FBBEV_DBP_QC;
        const long long base = (((long long)cam * B + b) * Q + (inb ? (long long)qy * gw + qx : 0)) * ZA;
        unsigned int mask4;
        __builtin_memcpy(&mask4, mask + base, 4);
        const fbbev_v4f r01 = *reinterpret_cast<const fbbev_v4f*>(ref_cam + base * 2);
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
        rec[3 * ZA] = (inb && mask4 != 0u) ? 1.f : 0.f;
    }
    const int m = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int qy = y0 + (lane >> plog), qx = x0 + (lane & (pw - 1));
    const bool valid = qy < gh && qx < gw;
    const long long bq = (long long)b * Q + (valid ? (long long)qy * gw + qx : 0);
    const float* my_qc = qc + (size_t)lane * FBBEV_DBP_QC;
    const int LP = L * P;
    auto unit_words = [&](int l) {                            // coalesced: item i = (query, point, head) in the head-minor layouts' order
        for (int i = threadIdx.x; i < 64 * P * MH; i += NT) {
            const int ql = i / (P * MH), r = i - ql * (P * MH), p = r / MH, h = r - p * MH;
            const int uy = y0 + (ql >> plog), ux = x0 + (ql & (pw - 1));
            if (!(uy < gh && ux < gw)) continue;
            const long long ubq = (long long)b * Q + (long long)uy * gw + ux, uu = ubq * MH + h;
            const long long io = (((head_minor & 1) ? ubq * LP * MH + h : uu * LP) + (long long)(l * P + p) * ((head_minor & 1) ? MH : 1)) * 2;
            const long long ia = ((head_minor & 2) ? ubq * LP * MH + h : uu * LP) + (long long)(l * P + p) * ((head_minor & 2) ? MH : 1);
            const fbbev_v2f t = *reinterpret_cast<const fbbev_v2f*>(offsets + io);
            float* so = off_s + ql * FBBEV_DBP_OQ + (p * MH + h) * 2;
            so[0] = t[0]; so[1] = t[1];
            att_s[ql * FBBEV_DBP_AQ + h * P + p] = attn[ia];
        }
    };
    unit_words(0);
    __syncthreads();                                                                        // records + level 0
    int count = 0;
    for (int cam = 0; cam < Ncam; ++cam) count += (fbbev_lds_ld_f32(my_qc + (size_t)cam * 64 * FBBEV_DBP_QC + 3 * ZA) != 0.f) ? 1 : 0;
    fbbev_v2f acc[DH / 2];
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) { acc[c][0] = 0.f; acc[c][1] = 0.f; }
    const float* my_off = off_s + lane * FBBEV_DBP_OQ + m * 2;
    const float* my_att = att_s + lane * FBBEV_DBP_AQ + m * P;
    const char* pb = reinterpret_cast<const char*>(planes);
    for (int l = 0; l < L; ++l) {
        const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
        const float fsh = (float)sh, fsw = (float)sw;
        const int lvl_off = (int)level_start[l] * DH;
        fbbev_v2f o[P];
        float a[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            o[p][0] = fbbev_lds_ld_f32(my_off + p * MH * 2); o[p][1] = fbbev_lds_ld_f32(my_off + p * MH * 2 + 1);
            a[p] = fbbev_lds_ld_f32(my_att + p);
        }
        if (l + 1 < L) {                                                                    // the next level's words arrive under this level's samples
            __syncthreads();
            unit_words(l + 1);
        }
        for (int cam = 0; cam < Ncam; ++cam) {
            const float* rec = my_qc + (size_t)cam * 64 * FBBEV_DBP_QC;
            const bool hit = valid && fbbev_lds_ld_f32(rec + 3 * ZA) != 0.f;
            if (__ballot(hit) == 0ull) continue;
            float rx[ZA], ry[ZA], dw[ZA];
#pragma unroll
            for (int z = 0; z < ZA; ++z) {
                rx[z] = fbbev_lds_ld_f32(rec + z); ry[z] = fbbev_lds_ld_f32(rec + ZA + z); dw[z] = fbbev_lds_ld_f32(rec + 2 * ZA + z);
            }
            const char* plane = pb + (((long long)b * Ncam + cam) * MH + m) * (long long)S * DH * 4;
            fbbev_daf_pending<DH> pend[2];
            auto start = [&](int p, fbbev_daf_pending<DH>& slot) {
                const int z = p % ZA;
                const float loc_w = rx[z] + __fdiv_rn(o[p][0], fsw), loc_h = ry[z] + __fdiv_rn(o[p][1], fsh);
                fbbev_daf_issue<DH>(plane, lvl_off, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, sh, sw, a[p] * dw[z], hit, slot);
            };
            start(0, pend[0]);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (p + 1 < P) start(p + 1, pend[(p + 1) & 1]);
                fbbev_sched_fence();
                fbbev_daf_consume<DH>(pend[p & 1], acc);
                fbbev_sched_fence();
            }
        }
        __syncthreads();                                                                    // the next level's words are in the tiles
    }
    if (!valid) return;
    const float inv = (float)(count > 1 ? count : 1);
    float* dst = slots + bq * (MH * DH) + m * DH;
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) {
        fbbev_v2f r;
        r[0] = acc[c][0] / inv; r[1] = acc[c][1] / inv;
        *reinterpret_cast<fbbev_v2f*>(dst + 2 * c) = r;
    }
}
