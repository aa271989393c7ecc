// da_kernels.h -- fused depth-aware spatial cross-attention sampling (backward projection).
//
// Replaces, in ONE launch, the sampling core of DA_SpatialCrossAttention + DA_MSDeformableAttention
// (fbbev/view_transformation/backward_projection/bevformer_utils/spatial_cross_attention_depth.py):
//   :163-169  6*B nonzero() host syncs building per-camera query lists, padded to max_len
//   :173-186  rebatch of queries / reference points / depths in Python loops (18*B index-puts)
//   :196-199  one-hot of the query depth bin  -> (B*6, L, Za, DC) int64
//   :584-590  MSDA #2: the whole DC-channel depth distribution sampled at every reference point,
//             then dotted with the one-hot                     -> here: ONE bilinear sample of the
//             query's own bin plane (identical value: the dot product with a one-hot selects it)
//   :592-595  attention_weights *= depth weight (no renormalisation); MSDA #3 value sampling
//   :208-216  scatter-add back per camera in camera order, divide by the number of cameras hit
// The per-query Linear layers (sampling_offsets, attention_weights) do not depend on the camera, so
// the host computes them ONCE per BEV query (the reference recomputes them per (camera, query) pair
// after rebatching) and hands them in.
//
// Semantics kept exactly: a query "hits" a camera if ANY of its Za anchors projects inside the image
// (per_cam_mask.sum(-1) > 0); for a hit camera ALL Za anchors are sampled, including the ones whose
// own mask bit is false; point index p = i*Za + z uses anchor z = p % Za (:560-570); cameras are
// accumulated in index order; empty hit set -> count clamped to 1.
#pragma once
#include "rt.h"
#include "msda_kernels.h"

#define FBBEV_DA_MAX_ZA 8
// LDS plane of the fixed-point backward kernels: word index of token t, and words of an n-token plane
#define FBBEV_DA_PLANE_IDX(t, HS) ((t) * (HS) + ((t) >> 3))
#define FBBEV_DA_PLANE_WORDS(n, HS) ((n) * (HS) + ((n) >> 3) + 1)

// One bilinear corner of a sample into a fixed-point LDS plane: DH adds of floor(w * tg[c] + 0.5) at dst[c], inside ONE divergent region.
template <int DH>
__device__ __forceinline__ void fbbev_lds_corner_add(bool live, long long* dst, float w, const float (&tg)[DH]) {
    if (!live) return;
    static_assert(DH % 2 == 0, "channel pairs");
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        const fbbev_v2f pr = fbbev_v2f{tg[c], tg[c + 1]} * w;                        // v_pk_mul_f32: two channels per instruction
        fbbev_lds_atomic_add_i64(dst + c, (long long)fbbev_cvt_rpi(pr[0]));
        fbbev_lds_atomic_add_i64(dst + c + 1, (long long)fbbev_cvt_rpi(pr[1]));
    }
}
#define FBBEV_DA_BWD_MAXP 8     // sampling points per level the split backward batches (FB-OCC: 8)

// bilinear sample of ONE plane (H,W) row-major at normalised (x,y), MSDA validity/padding rules
__device__ __forceinline__ float fbbev_plane_sample(const float* __restrict__ plane, int H, int W, float x,
                                                    float y) {
    const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return 0.f;
    const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, H, W, 1);
    const float v1 = s.o1 >= 0 ? plane[s.o1] : 0.f;
    const float v2 = s.o2 >= 0 ? plane[s.o2] : 0.f;
    const float v3 = s.o3 >= 0 ? plane[s.o3] : 0.f;
    const float v4 = s.o4 >= 0 ? plane[s.o4] : 0.f;
    return s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
}

// value (B*Ncam,S,M,Dh); pred_depth (B*Ncam,DC,H0,W0); ref_cam (Ncam,B,Q,Za,2); mask (Ncam,B,Q,Za) u8;
// qdepth (Ncam,B,Q,Za); offsets (B,Q,M,L,P,2) raw; attn (B,Q,M,L,P) softmaxed [head_minor: (B,Q,L,P,M,2) / (B,Q,L,P,M)];
// slots (B,Q,M*Dh)
__global__ void __launch_bounds__(256)
k_da_cross_attn_fwd(long long n, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                    const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                    const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                    const float* __restrict__ qdepth, const float* __restrict__ offsets,
                    const float* __restrict__ attn, int B, int Ncam, int S, int M, int Dh, int L, int Q, int P,
                    int Za, int DC, float d0, float dstep, int head_minor, int HS, float* __restrict__ slots) {
    const int row_stride = M * HS;               // HS = floats between two heads of a value row (>= Dh; padding ignored)
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Dh);
        const long long unit = idx / Dh;               // (b*Q + q)*M + m
        const int m = (int)(unit % M);
        const long long bq = unit / M;
        const int q = (int)(bq % Q);
        const int b = (int)(bq / Q);
        float acc = 0.f;
        int count = 0;
        for (int cam = 0; cam < Ncam; ++cam) {
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            bool hit = false;
            for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            if (!hit) continue;
            ++count;
            const long long bn = (long long)b * Ncam + cam;
            float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA];
            for (int z = 0; z < Za; ++z) {
                rx[z] = ref_cam[(base + z) * 2];
                ry[z] = ref_cam[(base + z) * 2 + 1];
                // :196-197  bin = clip(floor((d - dbound[0]) / dbound[2]), 0, DC-1)
                float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                const int bin = (int)fb;
                dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + bin) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
            }
            float col = 0.f;
            for (int l = 0; l < L; ++l) {
                const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                const float* vp = value + (bn * S + level_start[l]) * row_stride +
                                  ((head_minor & 4) ? (c >> 2) * (M * 4) + m * 4 + (c & 3) : m * HS + c);
                for (int p = 0; p < P; ++p) {
                    // offsets / attn: (B,Q,M,L,P[,2]) as the Linear layers emit them, or head-minor (B,Q,L,P,M[,2]):
                    // head_minor bit 0 -> offsets, bit 1 -> attn; bit 2 -> quad-interleaved value rows (see the unit kernel)
                    const long long wm = (unit * L + l) * P + p, wh = ((bq * L + l) * P + p) * M + m;
                    const long long wo = (head_minor & 1) ? wh : wm, wa = (head_minor & 2) ? wh : wm;
                    const int z = p % Za;
                    const float loc_w = rx[z] + __fdiv_rn(offsets[wo * 2], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(offsets[wo * 2 + 1], (float)sh);
                    const float weight = attn[wa] * dw[z];
                    const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw) {
                        const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, row_stride);
                        const float v1 = s.o1 >= 0 ? vp[s.o1] : 0.f;
                        const float v2 = s.o2 >= 0 ? vp[s.o2] : 0.f;
                        const float v3 = s.o3 >= 0 ? vp[s.o3] : 0.f;
                        const float v4 = s.o4 >= 0 ? vp[s.o4] : 0.f;
                        col += (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * weight;
                    }
                }
            }
            acc += col;
        }
        slots[idx] = acc / (float)(count > 1 ? count : 1);
    }
}


// ---------------------------------------------------------------- unit-per-lane variant (the one FB-OCC shapes take)
// Same arithmetic, expression for expression, as k_da_cross_attn_fwd above (=> identical bits), different ownership: a
// lane owns a whole (b, q, head) unit -- all DH channels.  The camera hit test, the Za depth weights, the sampling
// offsets / attention weights and the bilinear setup are evaluated ONCE per unit instead of once per channel lane, and
// each corner is read as DH/2 eight-byte loads (a head's DH floats are contiguous; neighbouring lanes = neighbouring
// heads read one contiguous M*DH*4-byte row).  The channel-per-lane kernel spends its time in the vector L1's per-lane
// dword rate (7 loads per sample per lane, 10 lanes per unit at Dh = 10); this one issues 23 loads per sample per unit.
// WIDE: the head stride HS is a multiple of 4 floats and covers DH rounded up to 4 (the host pads value_proj's output
// rows, e.g. Dh = 10 -> HS = 12): every head chunk is 16-byte aligned and a corner is read as DHP/4 dwordx4 loads
// (3 L1 accesses instead of 5 eight-byte ones; the padding floats are loaded and ignored).
// ET: element type of the token rows -- 0 f32; 1 bf16 / 2 f16 (inference option, QI only: [chunk][head][8 elements]).
template <int DH, bool WIDE, bool QI, int ET = 0>
__global__ void __launch_bounds__(256)
k_da_cross_attn_fwd_unit(long long n_units, const void* __restrict__ value_, const int64_t* __restrict__ spatial_shapes,
                         const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                         const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                         const float* __restrict__ qdepth, const float* __restrict__ offsets,
                         const float* __restrict__ attn, int B, int Ncam, int S, int M, int L, int Q, int P, int Za,
                         int DC, float d0, float dstep, int head_minor, int HS, int stage_attn,
                         float* __restrict__ slots) {
    static_assert(DH % 2 == 0, "eight-byte loads");
    static_assert(!QI || WIDE, "quad-interleaved rows are read with 16-byte loads");
    static_assert(ET == 0 || QI, "16-bit rows are chunk-major");
    const float* value = static_cast<const float*>(value_);
    constexpr int CE = ET == 0 ? 4 : 8;           // elements per (chunk, head) piece: 16 bytes either way
    constexpr unsigned ESZ = ET == 0 ? 4u : 2u;
    // QI: a camera token's row is stored [chunk k][head m][4 floats] instead of [head m][HS floats]: the 8 head lanes
    // of a query then read ONE contiguous M*16-byte piece per load instruction (one 128-byte line at M = 8) where the
    // head-major row makes every one of the DHP/4 loads touch all of the row's lines.  Same floats, same arithmetic.
    const int head_off_m = QI ? CE : HS, chunk_stride = QI ? M * CE : 4;
    const int row_stride = M * HS;
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int LP = L * P, LDW = LP + 1;
    float* staged = fbbev_dyn_lds_f32();          // [256][LP+1] when stage_attn: the workgroup's attention weights
    // XCD-contiguous order: workgroup w runs on XCD w % 8 (each with its own L2); XCD x takes the x-th eighth of the
    // unit range -- a contiguous piece of the BEV plane, whose queries project into the same few camera regions -- instead
    // of every eighth workgroup of the whole plane (gridDim.x is a multiple of 8)
    const long long n_wg = (n_units + blockDim.x - 1) / blockDim.x, per_xcd = (n_wg + 7) / 8;
    for (long long w = blockIdx.x; (w >> 3) < per_xcd; w += gridDim.x) {
        const long long ubase = ((w & 7) * per_xcd + (w >> 3)) * blockDim.x;
        const long long unit = ubase + threadIdx.x;
        if (stage_attn) {
            // attn (B,Q,M,L,P): the workgroup's 256 units own 256*LP CONTIGUOUS floats; read them with coalesced
            // 16-byte loads once (a lane reading its own weights sample by sample touches a different 128-byte line
            // than each of its 63 neighbours, 32 times per camera) and hand them out from LDS, row pitch LP+1.
            __syncthreads();
            const long long rem = n_units - ubase;       // <= 0 for the padding workgroups of the last XCD
            const int nfl = rem <= 0 ? 0 : (int)(rem < (long long)blockDim.x ? rem : (long long)blockDim.x) * LP;
            const float* src = attn + ubase * LP;
            for (int i = threadIdx.x * 4; i < nfl; i += blockDim.x * 4) {
                const fbbev_v4f a = *reinterpret_cast<const fbbev_v4f*>(src + i);
                float* d = staged + (i / LP) * LDW + (i % LP);
                d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3];
            }
            __syncthreads();
        }
        if (unit >= n_units) continue;
        const float* my_attn = staged + threadIdx.x * LDW;
        const int m = (int)(unit % M);
        const long long bq = unit / M;
        const int q = (int)(bq % Q);
        const int b = (int)(bq / Q);
        float acc[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] = 0.f;
        // offsets of sample lp: (B,Q,M,L,P,2) -> unit*LP + lp, head-minor (B,Q,L,P,M,2) -> (bq*LP + lp)*M + m
        const fbbev_v2f* op = reinterpret_cast<const fbbev_v2f*>(offsets) + ((head_minor & 1) ? bq * LP * M + m : unit * LP);
        const int wo_step = (head_minor & 1) ? M : 1;
        int count = 0;
        for (int cam = 0; cam < Ncam; ++cam) {
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            bool hit = false;
            for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            if (!hit) continue;
            ++count;
            const long long bn = (long long)b * Ncam + cam;
            float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA];
            for (int z = 0; z < Za; ++z) {
                rx[z] = ref_cam[(base + z) * 2];
                ry[z] = ref_cam[(base + z) * 2 + 1];
                float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                const int bin = (int)fb;
                dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + bin) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
            }
            float col[DH];
#pragma unroll
            for (int c = 0; c < DH; ++c) col[c] = 0.f;
            fbbev_v2f o_next = op[0];
            int lp = 0;
            for (int l = 0; l < L; ++l) {
                const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                const unsigned lane_off = (unsigned)(((bn * S + level_start[l]) * row_stride + m * head_off_m) * ESZ);
                for (int p = 0; p < P; ++p, ++lp) {
                    // head-minor (B,Q,L,P,M[,2]): the 8 heads of a query read 64 contiguous bytes per sample; the
                    // (B,Q,M,L,P[,2]) layout strides lanes by L*P*8 bytes and thrashes the vector L1
                    // (head_minor bit 0 -> offsets, bit 1 -> attn; attn may stay in the layout the fast last-dim
                    // softmax produces: it is then staged through LDS above).  The next sample's offsets are requested
                    // before this sample's value loads: one exposed memory latency per sample instead of two.
                    const fbbev_v2f o = o_next;
                    o_next = op[(long long)(lp + 1 < LP ? lp + 1 : lp) * wo_step];
                    const int z = p % Za;
                    const float loc_w = rx[z] + __fdiv_rn(o[0], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(o[1], (float)sh);
                    // the staged weight through an explicit LDS pointer: as `stage_attn ? my_attn[lp] : attn[..]` the two
                    // (generic) POINTERS were selected and one flat_load_dword per sample issued -- generic address space,
                    // counted on both wait counters.  The value is first needed when the sample is blended, so the ds_read
                    // overlaps the corner loads.
                    float a;
                    if (stage_attn) a = fbbev_lds_ld_f32(my_attn + lp);
                    else a = attn[(head_minor & 2) ? (bq * LP + lp) * M + m : unit * LP + lp];
                    const float weight = a * dw[z];
                    const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw) {
                        const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, row_stride);
                        if constexpr (ET == 0) fbbev_unit_sample<DH, WIDE ? 4 : 2>(value, lane_off, s, chunk_stride, weight, col);
                        else fbbev_unit_sample16<DH, ET>(value_, lane_off, s, chunk_stride, weight, col);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < DH; ++c) acc[c] += col[c];
        }
        const float inv = (float)(count > 1 ? count : 1);
        float* dst = slots + unit * DH;
#pragma unroll
        for (int c = 0; c < DH; c += 2) {
            fbbev_v2f r;
            r[0] = acc[c] / inv; r[1] = acc[c + 1] / inv;
            *reinterpret_cast<fbbev_v2f*>(dst + c) = r;
        }
    }
}


// ---------------------------------------------------------------- pipelined unit-per-lane forward (round 3)
// The loop above drains every sample's 12 corner loads before it blends them (ISA: `gload x13 | vmcnt(0) | valu x98`), so
// only the 3 waves per SIMD overlap anything; SQ counters at BASELINE configs[2]: waves parked 42 %, issue-stalled 34 %
// (profiles/r03_pmc_fb_BL3_B4_before.json).  This variant keeps TWO samples in flight per lane: the corner loads of
// sample i+1 are issued before sample i is blended (tools/micro/unit_sampler_pipeline.hip on the GPU: 1.35 -> 1.06 ms,
// profiles/r03_exp_unit_sampler_pipeline.jsonl).  What makes that possible, all visible in the ISA skeleton
// (tests/test_kernel_resources.py guards it):
//   * no branch between issue and consume: an out-of-image sample or a zero-padded corner does not skip its loads, it
//     reads the ZERO TOKEN the host appends to the value buffer (token index B*Ncam*S, all +0.0f) -- the padded corner
//     then contributes w * 0 exactly as in the reference (no 40 v_cndmask per sample either), an out-of-image sample
//     gets weight 0;
//   * two register slots addressed at compile time (the loop is unrolled over the ZA anchors: ZA samples = ZA/2 slot
//     pairs; the anchor's reference point and depth weight are then static registers instead of a dynamically indexed
//     array), the look-ahead issue past the last sample is a dummy on the zero token, never a branch;
//   * the third chunk of a Dh = 10 head (channels 8, 9 + two padding floats) is read as 8 bytes: the two dead floats of a
//     16-byte load are destination registers the allocator reuses while the load is in flight, and the write-after-write
//     wait drains the queue; also 160 instead of 192 bytes per sample through the vector L1;
//   * the offsets of the NEXT group of ZA samples are requested first in a group's body: `vmcnt` retires in order, so they
//     are older than every corner load issued after them and never waited past.
// Arithmetic: the blend is the reference's `(w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight` per channel, on channel pairs
// (v_pk_mul_f32 / v_pk_fma_f32); `offset / size` is the correctly rounded division of the unit kernel and the backward kernels
// (round 3 used offset * (1 / size): one ulp of a sub-pixel offset, but inference and training forward then disagreed in bits).
// Preconditions (launcher): chunk-major fp32 rows (QI), head-minor offsets, attention weights staged through LDS,
// P % ZA == 0, ZA even, DH in {8, 10}.
template <int DH>
struct fbbev_da_pending {
    static constexpr int NF = DH / 4;                    // full 16-byte chunks of a head
    fbbev_v4f a1[NF], a2[NF], a3[NF], a4[NF];
    fbbev_v2f t1, t2, t3, t4;                            // 8-byte tail chunk (DH % 4 == 2)
    float w1, w2, w3, w4, weight;
};

template <int DH>
__device__ __forceinline__ void fbbev_da_issue(const char* __restrict__ vb, unsigned lane_off, unsigned zero_off, unsigned cs,
                                               int row_stride, float h_im, float w_im, int sh, int sw, float weight,
                                               bool enable, fbbev_da_pending<DH>& p) {
    const bool live = enable && h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw;
    // an out-of-image sample is set up at (0, 0): finite weights, every corner on the zero token, weight 0
    const fbbev_bilinear s = fbbev_bilinear_setup(live ? h_im : 0.f, live ? w_im : 0.f, sh, sw, row_stride);
    p.w1 = s.w1; p.w2 = s.w2; p.w3 = s.w3; p.w4 = s.w4;
    p.weight = live ? weight : 0.f;
    const unsigned b1 = (live && s.o1 >= 0) ? lane_off + (unsigned)s.o1 * 4u : zero_off;
    const unsigned b2 = (live && s.o2 >= 0) ? lane_off + (unsigned)s.o2 * 4u : zero_off;
    const unsigned b3 = (live && s.o3 >= 0) ? lane_off + (unsigned)s.o3 * 4u : zero_off;
    const unsigned b4 = (live && s.o4 >= 0) ? lane_off + (unsigned)s.o4 * 4u : zero_off;
    constexpr int NF = DH / 4;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        p.a1[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b1 + k * cs));
        p.a2[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b2 + k * cs));
        p.a3[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b3 + k * cs));
        p.a4[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b4 + k * cs));
    }
    if constexpr (DH % 4 == 2) {
        p.t1 = *reinterpret_cast<const fbbev_v2f*>(vb + (b1 + NF * cs));
        p.t2 = *reinterpret_cast<const fbbev_v2f*>(vb + (b2 + NF * cs));
        p.t3 = *reinterpret_cast<const fbbev_v2f*>(vb + (b3 + NF * cs));
        p.t4 = *reinterpret_cast<const fbbev_v2f*>(vb + (b4 + NF * cs));
    }
}

__device__ __forceinline__ fbbev_v2f fbbev_pair(const fbbev_v4f& a, int hi) {
    fbbev_v2f r;
    r[0] = a[2 * hi]; r[1] = a[2 * hi + 1];
    return r;
}

template <int DH>
__device__ __forceinline__ void fbbev_da_consume(const fbbev_da_pending<DH>& p, fbbev_v2f (&col)[DH / 2]) {
    constexpr int NF = DH / 4;
#pragma unroll
    for (int k = 0; k < NF; ++k)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const fbbev_v2f v1 = fbbev_pair(p.a1[k], hi), v2 = fbbev_pair(p.a2[k], hi);
            const fbbev_v2f v3 = fbbev_pair(p.a3[k], hi), v4 = fbbev_pair(p.a4[k], hi);
            col[2 * k + hi] += (p.w1 * v1 + p.w2 * v2 + p.w3 * v3 + p.w4 * v4) * p.weight;
        }
    if constexpr (DH % 4 == 2) col[DH / 2 - 1] += (p.w1 * p.t1 + p.w2 * p.t2 + p.w3 * p.t3 + p.w4 * p.t4) * p.weight;
    // the blend happens HERE, before the next sample's loads are issued (see fbbev_pin)
#pragma unroll
    for (int c = 0; c < DH / 2; ++c) fbbev_pin(col[c]);
}

// WPS: waves per SIMD the register allocation is bounded for (3 -> 168 VGPRs: a dozen values of the per-camera prologue
// spill, none inside the sample loop; 2 -> the natural 183)
template <int DH, int ZA, int WPS>
__global__ void __launch_bounds__(256, WPS)
k_da_cross_attn_fwd_pipe(long long n_units, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                         const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                         const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                         const float* __restrict__ qdepth, const float* __restrict__ offsets,
                         const float* __restrict__ attn, int B, int Ncam, int S, int M, int L, int Q, int P,
                         int DC, float d0, float dstep, int HS, unsigned zero_token_bytes, int attn_logits, int bev_w,
                         float* __restrict__ slots) {
    static_assert(DH % 2 == 0 && (DH % 4 == 0 || DH % 4 == 2), "channel pairs");
    static_assert(ZA % 2 == 0 && ZA <= FBBEV_DA_MAX_ZA, "two register slots alternate over the anchors");
    const char* vb = reinterpret_cast<const char*>(value);
    const int row_stride = M * HS;                // floats per token
    const unsigned cs = (unsigned)M * 16u;        // bytes between the chunks of a head (chunk-major rows)
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int LP = L * P, LDW = LP + 1, gpl = P / ZA;
    float* staged = fbbev_dyn_lds_f32();          // [256][LP+1]: the workgroup's attention weights
    // Which 256 (query, head) units a workgroup owns, and which 64 a wave:
    //   bev_w == 0 : 32 consecutive queries x 8 heads in unit order (any M, any Q);
    //   bev_w > 0  : (M == 8, Q = bev_h x bev_w) the 8 heads of an 8 (x) x 4 (y) PATCH of the BEV grid, and a wave = 4 heads
    //                of a 4 x 4 sub-patch.  The vector L1 of this kernel is saturated in ACCESSES (TCP_TOTAL_CACHE_ACCESSES
    //                per CU ~ the kernel's cycles, profiles/r03_pmc_fb_BL3_B4_before.json): the 8 heads of a query sample 8
    //                different tokens (the offsets are per head), while the neighbours of a query IN BOTH BEV directions
    //                sample (nearly) the same ones for a given head -- a 4 x 4 patch of 4 heads touches fewer distinct
    //                lines per load instruction than 8 queries of a row x 8 heads (tools/micro/unit_sampler_pipeline.hip:
    //                0.92 -> 0.69 ms, profiles/r03_exp_unit_sampler_patch.jsonl).  Every tensor keeps its layout; only the
    //                lane -> unit map changes, so the results are the same bits.
    const bool patch = bev_w > 0;
    const int bev_h = patch ? Q / bev_w : 0;
    const int pxn = patch ? (bev_w + 7) / 8 : 0, pyn = patch ? (bev_h + 3) / 4 : 0;
    const long long n_wg = patch ? (long long)B * pxn * pyn : (n_units + blockDim.x - 1) / blockDim.x;
    const long long per_xcd = (n_wg + 7) / 8;
    for (long long w = blockIdx.x; (w >> 3) < per_xcd; w += gridDim.x) {
        const long long wgid = (w & 7) * per_xcd + (w >> 3);                          // XCD-contiguous workgroup order
        long long unit = -1;
        int lu = threadIdx.x;                                                         // row of the LDS weight table
        __syncthreads();
        if (!patch) {
            const long long ubase = wgid * blockDim.x;
            unit = ubase + threadIdx.x;
            const long long rem = n_units - ubase;
            const int nfl = rem <= 0 ? 0 : (int)(rem < (long long)blockDim.x ? rem : (long long)blockDim.x) * LP;
            const float* src = attn + ubase * LP;
            for (int i = threadIdx.x * 4; i < nfl; i += blockDim.x * 4) {
                const fbbev_v4f a = *reinterpret_cast<const fbbev_v4f*>(src + i);
                float* d = staged + (i / LP) * LDW + (i % LP);
                d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3];
            }
            if (unit >= n_units) unit = -1;
        } else if (wgid < n_wg) {
            const int pb = (int)(wgid / ((long long)pxn * pyn)), pi = (int)(wgid - (long long)pb * pxn * pyn);
            const int py = pi / pxn, px = pi - py * pxn;
            const int x0 = px * 8, nx = bev_w - x0 < 8 ? bev_w - x0 : 8;
            // the patch's weights: 4 row segments of nx queries x 8 heads x LP floats, each contiguous in (B,Q,M,L,P)
            for (int r = 0; r < 4; ++r) {
                const int y = py * 4 + r;
                if (y >= bev_h) break;
                const float* src = attn + (((long long)pb * Q + (long long)y * bev_w + x0) * M) * LP;
                const int nfl = nx * M * LP;
                for (int i = threadIdx.x * 4; i < nfl; i += blockDim.x * 4) {
                    const fbbev_v4f a = *reinterpret_cast<const fbbev_v4f*>(src + i);
                    float* d = staged + (r * 8 * M + i / LP) * LDW + (i % LP);
                    d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3];
                }
            }
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const int mh = 4 * (wave & 1) + (lane & 3), qi = lane >> 2;
            const int xi = (wave >> 1) * 4 + (qi & 3), r = qi >> 2;
            const int y = py * 4 + r, x = x0 + xi;
            lu = (r * 8 + xi) * M + mh;
            if (y < bev_h && x < bev_w) unit = ((long long)pb * Q + (long long)y * bev_w + x) * M + mh;
        }
        __syncthreads();
        if (unit < 0) continue;
        float* my_attn = staged + lu * LDW;
        if (attn_logits) {
            // `attn` holds the raw output of the attention_weights Linear: the softmax over the unit's L*P weights
            // (spatial_cross_attention_depth.py:546-551) runs here on the staged row -- one separate softmax launch and one
            // write + read of the (B,Q,M,L*P) tensor less; exp(x - max) / sum like ATen's kernel (v_exp_f32 based)
            float mx = my_attn[0];
            for (int i = 1; i < LP; ++i) mx = fmaxf(mx, my_attn[i]);
            float sum = 0.f;
            for (int i = 0; i < LP; ++i) { const float e = __expf(my_attn[i] - mx); my_attn[i] = e; sum += e; }
            const float inv_sum = 1.f / sum;
            for (int i = 0; i < LP; ++i) my_attn[i] *= inv_sum;
        }
        const int m = (int)(unit % M);
        const long long bq = unit / M;
        const int q = (int)(bq % Q);
        const int b = (int)(bq / Q);
        const unsigned zero_off = zero_token_bytes + (unsigned)m * 16u;
        fbbev_v2f acc[DH / 2];
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) { acc[c][0] = 0.f; acc[c][1] = 0.f; }
        const fbbev_v2f* op = reinterpret_cast<const fbbev_v2f*>(offsets) + bq * LP * M + m;   // head-minor (B,Q,L,P,M,2)
        // hit test of all cameras first (independent loads in flight together): a query hits a camera if ANY anchor does
        unsigned hits = 0u;
        for (int cam = 0; cam < Ncam; ++cam) {
            const long long base = (((long long)cam * B + b) * Q + q) * ZA;
            bool hit = false;
#pragma unroll
            for (int z = 0; z < ZA; ++z) hit = hit || (mask[base + z] != 0);
            hits |= hit ? (1u << cam) : 0u;
        }
        int count = 0;
        for (int cam = 0; cam < Ncam; ++cam) {
            if (!((hits >> cam) & 1u)) continue;
            ++count;
            const long long base = (((long long)cam * B + b) * Q + q) * ZA;
            const long long bn = (long long)b * Ncam + cam;
            // offsets of samples 0 and 1: requested before the depth weights are evaluated
            fbbev_v2f o_a = op[0], o_b = op[M];
            float rx[ZA], ry[ZA], dw[ZA];
#pragma unroll
            for (int z = 0; z < ZA; ++z) {
                rx[z] = ref_cam[(base + z) * 2];
                ry[z] = ref_cam[(base + z) * 2 + 1];
                float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                const int bin = (int)fb;
                dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + bin) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
            }
            fbbev_v2f col[DH / 2];
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) { col[c][0] = 0.f; col[c][1] = 0.f; }
            const long long cam_tok = bn * S;
            // issue-stream state: sizes / reciprocals / this lane's byte offset of the level the NEXT issue samples
            int sh = H0, sw = W0;
            float fsh = (float)sh, fsw = (float)sw;
            unsigned lane_off = (unsigned)(((cam_tok + level_start[0]) * row_stride + m * 4) * 4);
            fbbev_da_pending<DH> pa, pb;
            // issue sample lp (anchor z) with the offsets `o`; the offsets of sample lp + 2 -- the next user of the same
            // register pair -- are requested FIRST: `vmcnt` retires in order, so that small load is older than the corner
            // loads issued behind it and is complete whenever they are
            auto start = [&](int lp, int z, bool enable, fbbev_v2f& o, fbbev_da_pending<DH>& slot) {
                const float loc_w = rx[z] + __fdiv_rn(o[0], fsw), loc_h = ry[z] + __fdiv_rn(o[1], fsh);   // the division of every other DA kernel
                const int nx = lp + 2 < LP ? lp + 2 : LP - 1;          // clamped: past the end a duplicate nobody uses
                o = op[(long long)nx * M];
                const float weight = fbbev_lds_ld_f32(my_attn + lp) * dw[z];
                fbbev_da_issue<DH>(vb, lane_off, zero_off, cs, row_stride, loc_h * fsh - 0.5f, loc_w * fsw - 0.5f, sh, sw,
                                   weight, enable, slot);
            };
            start(0, 0, true, o_a, pa);
            int g = 0;                             // flat group index = lp / ZA
            for (int l = 0; l < L; ++l) {
                // the level after this one (scalar loads, once per level, outside the pipelined loop); the last level looks
                // ahead into itself and its look-ahead sample is disabled
                const int ln = l + 1 < L ? l + 1 : l;
                const int nsh = (int)spatial_shapes[2 * ln], nsw = (int)spatial_shapes[2 * ln + 1];
                const float nfsh = (float)nsh, nfsw = (float)nsw;
                const unsigned nlane_off = (unsigned)(((cam_tok + level_start[ln]) * row_stride + m * 4) * 4);
                for (int gl = 0; gl < gpl; ++gl, ++g) {
                    const bool last_of_level = gl + 1 == gpl;
                    const bool more = !(last_of_level && l + 1 == L);          // a group follows this one
                    const int gn = more ? g + 1 : g;
#pragma unroll
                    for (int z = 0; z < ZA; z += 2) {
                        start(g * ZA + z + 1, z + 1, true, o_b, pb);   // corners of sample z+1 ...
                        fbbev_sched_fence();
                        fbbev_da_consume<DH>(pa, col);                 // ... in flight while sample z is blended
                        fbbev_sched_fence();
                        if (z + 2 < ZA) {
                            start(g * ZA + z + 2, z + 2, true, o_a, pa);
                        } else {
                            // look-ahead into the next group: selects, not branches (a branch with memory operations on
                            // one side makes the wait-count pass assume the worst at the join)
                            sh = last_of_level ? nsh : sh; sw = last_of_level ? nsw : sw;
                            fsh = last_of_level ? nfsh : fsh; fsw = last_of_level ? nfsw : fsw;
                            lane_off = last_of_level ? nlane_off : lane_off;
                            start(gn * ZA, 0, more, o_a, pa);         // past the last sample: a dummy on the zero token
                        }
                        fbbev_sched_fence();
                        fbbev_da_consume<DH>(pb, col);
                        fbbev_sched_fence();
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) acc[c] += col[c];
        }
        const float inv = (float)(count > 1 ? count : 1);
        float* dst = slots + unit * DH;
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) {
            fbbev_v2f r;
            r[0] = acc[c][0] / inv; r[1] = acc[c][1] / inv;
            *reinterpret_cast<fbbev_v2f*>(dst + 2 * c) = r;
        }
    }
}


// ---------------------------------------------------------------- backward of the fused sampling (training path)
// Replaces the autograd chain of the reference's training step through DA_SpatialCrossAttention /
// DA_MSDeformableAttention (spatial_cross_attention_depth.py:163-216,513-595 -> two MultiScaleDeformableAttnFunction
// backward launches, multi_scale_deformable_attn_function.py:137-172, the one-hot / rebatch / scatter index ops and
// their 6*B host syncs) with one launch.  A group of GW lanes owns a (b,q,head) unit, lane = channel (as k_msda_bwd):
// the value-gradient atomics of a corner then hit Dh CONSECUTIVE floats from consecutive lanes (one coalesced atomic
// request per corner instead of Dh scattered ones -- the unit-per-lane form of this kernel was 3x slower for it).
//   grad_attn, grad_offsets : owned by the unit -> lane 0 of the group accumulates them over the hit cameras with plain
//                             read-modify-writes (buffers pre-zeroed by the caller; layouts = the forward's)
//   grad_value              : fp32 hardware atomics (corners shared between units), as mmcv's col2im
//   grad_pred_depth         : the depth weight dw[z] is ONE bilinear sample of the query's bin plane; its gradient
//                             ddw[z] = sum over the samples of anchor z of attn * <grad, sampled value> goes back
//                             to the four corners of that plane by atomics
// Bilinear-gradient terms follow mmcv's ms_deform_attn_col2im_bilinear (grad_h_weight / grad_w_weight); since
// loc = ref + offset / size and im = loc * size - 0.5, d im / d offset = 1.
// Control flow is wave-uniform (cameras nobody in the wave hits are skipped by a ballot) because the group
// reductions are cross-lane shuffles.
template <int GW>
__global__ void __launch_bounds__(256)
k_da_cross_attn_bwd(long long n_units, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                    const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                    const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                    const float* __restrict__ qdepth, const float* __restrict__ offsets,
                    const float* __restrict__ attn, const float* __restrict__ grad_slots, int B, int Ncam, int S,
                    int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor, int HS,
                    float* __restrict__ grad_value, float* __restrict__ grad_pred_depth,
                    float* __restrict__ grad_offsets, float* __restrict__ grad_attn) {
    const int slot = threadIdx.x % GW;
    const long long unit = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / GW;
    const bool active = unit < n_units;
    const long long u = active ? unit : 0;
    const int row_stride = M * HS;               // value / grad_value rows: M heads of HS floats (Dh used)
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int m = (int)(u % M);
    const long long bq = u / M;
    const int q = (int)(bq % Q);
    const int b = (int)(bq / Q);
    const bool chan = active && slot < Dh;
    int count = 0;
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long base = (((long long)cam * B + b) * Q + q) * Za;
        bool hit = false;
        for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
        count += hit ? 1 : 0;
    }
    const float g = chan ? grad_slots[u * Dh + slot] / (float)(count > 1 ? count : 1) : 0.f;
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long base = (((long long)cam * B + b) * Q + q) * Za;
        bool hit = false;
        for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
        hit = hit && active;
        if (__ballot(hit ? 1 : 0) == 0ull) continue;                      // wave-uniform skip
        const long long bn = (long long)b * Ncam + cam;
        float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA], ddw[FBBEV_DA_MAX_ZA];
        int bin[FBBEV_DA_MAX_ZA];
        for (int z = 0; z < Za; ++z) {
            rx[z] = ry[z] = dw[z] = ddw[z] = 0.f;
            bin[z] = 0;
            if (hit) {
                rx[z] = ref_cam[(base + z) * 2];
                ry[z] = ref_cam[(base + z) * 2 + 1];
                float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                bin[z] = (int)fb;
                dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + bin[z]) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
            }
        }
        for (int l = 0; l < L; ++l) {
            const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
            const long long voff = (bn * S + level_start[l]) * row_stride +
                                   ((head_minor & 4) ? (slot >> 2) * (M * 4) + m * 4 + (slot & 3) : m * HS + slot);
            for (int p = 0; p < P; ++p) {
                const long long wm = (u * L + l) * P + p, wh = ((bq * L + l) * P + p) * M + m;
                const long long wo = (head_minor & 1) ? wh : wm, wa = (head_minor & 2) ? wh : wm;
                const int z = p % Za;
                float a = 0.f, h_im = -2.f, w_im = -2.f;
                if (hit) {
                    const float loc_w = rx[z] + __fdiv_rn(offsets[wo * 2], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(offsets[wo * 2 + 1], (float)sh);
                    a = attn[wa];
                    h_im = loc_h * sh - 0.5f;
                    w_im = loc_w * sw - 0.5f;
                }
                const bool inr = hit && h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw;
                const float weight = a * dw[z];
                float dot = 0.f, gx = 0.f, gy = 0.f;     // <g, sample>, d/d w_im, d/d h_im of <g, sample>
                if (inr && chan) {
                    const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, row_stride);
                    const float* vp = value + voff;
                    float* gp = grad_value + voff;
                    const float tgv = g * weight;
                    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                    if (s.o1 >= 0) { v1 = vp[s.o1]; fbbev_atomic_add_f32(gp + s.o1, s.w1 * tgv); }
                    if (s.o2 >= 0) { v2 = vp[s.o2]; fbbev_atomic_add_f32(gp + s.o2, s.w2 * tgv); }
                    if (s.o3 >= 0) { v3 = vp[s.o3]; fbbev_atomic_add_f32(gp + s.o3, s.w3 * tgv); }
                    if (s.o4 >= 0) { v4 = vp[s.o4]; fbbev_atomic_add_f32(gp + s.o4, s.w4 * tgv); }
                    dot = g * (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4);
                    gy = g * (-s.hw * v1 - s.lw * v2 + s.hw * v3 + s.lw * v4);
                    gx = g * (-s.hh * v1 + s.hh * v2 - s.lh * v3 + s.lh * v4);
                }
                dot = fbbev_group_sum<GW>(dot);
                gx = fbbev_group_sum<GW>(gx);
                gy = fbbev_group_sum<GW>(gy);
                if (inr && slot == 0) {
                    grad_attn[wa] += dw[z] * dot;
                    grad_offsets[wo * 2] += weight * gx;
                    grad_offsets[wo * 2 + 1] += weight * gy;
                    ddw[z] += a * dot;
                }
            }
        }
        if (hit && slot == 0) {
            for (int z = 0; z < Za; ++z) {       // dw[z] -> the four corners of the query's bin plane (fbbev_plane_sample)
                const float h_im = ry[z] * H0 - 0.5f, w_im = rx[z] * W0 - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H0 && w_im < (float)W0) || ddw[z] == 0.f) continue;
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, H0, W0, 1);
                float* gd = grad_pred_depth + (bn * DC + bin[z]) * (long long)(H0 * W0);
                if (s.o1 >= 0) fbbev_atomic_add_f32(gd + s.o1, s.w1 * ddw[z]);
                if (s.o2 >= 0) fbbev_atomic_add_f32(gd + s.o2, s.w2 * ddw[z]);
                if (s.o3 >= 0) fbbev_atomic_add_f32(gd + s.o3, s.w3 * ddw[z]);
                if (s.o4 >= 0) fbbev_atomic_add_f32(gd + s.o4, s.w4 * ddw[z]);
            }
        }
    }
}


// ---------------------------------------------------------------- backward with LDS-resident value-gradient planes
// k_da_cross_attn_bwd sends every corner of every sample to the value gradient with a global fp32 atomic: at the shipped
// shapes (Q = 10^4, one 16x44 level) ~110 adds land on each of the 1.6 M gradient floats, across all 8 XCDs -- 1.5 ms at
// B = 4 (14.2 ms at the configs[2] pyramid), bound by the atomic rate.  The replacement (fbbev_da_cross_attn_bwd_ws) keeps
// a head's gradient plane in LDS, accumulates with LDS atomics and writes it with plain 16-byte stores to the workgroup's
// slice of a partial buffer  part[b][m][chunk][cam][S*HS];  k_da_bwd_reduce then sums the chunks into grad_value in the
// layout of `value` -- no global atomic touches the value gradient.
// The plane is FIXED POINT: 64-bit integers in units of 2^-30 of the power of two above max|grad_slots| of the chunk.
// ds_add_f32 retires ~0.8 lanes per ns and CU on gfx950, ds_add_u64 22 (profiles/r02_micro_lds_atomics.jsonl: the fp32
// LDS atomic is 40x slower than the integer ones), and integer adds commute: the value gradient is bit-reproducible run
// to run, which neither the fp32-atomic kernels here nor mmcv's col2im are.  A contribution w*g*attn*dw is at most
// max|g| in magnitude, so it is rounded ONCE to a multiple of 2^-30 of that bound (finer than its own fp32 ulp for
// everything within 2^-6 of the largest contribution) and the <= q_per_chunk*L*P adds of a plane cannot overflow 63 bits;
// the plane is converted back with one rounding.  Non-finite upstream gradients turn the chunk's planes into NaN.
// History of the design (profiles/r02_da_bwd_variants.jsonl): a first kernel did everything in one loop -- four lanes per
// unit, value loads, LDS adds, group shuffles, staged read-modify-writes, depth atomics: 0.39 ms at the shipped shape, of
// which 0.3 ms were the 8.7 M scattered fp32 global atomics of the depth-distribution gradient (8 head workgroups each
// adding the same corners).  It was replaced by the two kernels below.
// Split by what the gradients need (the forward kernel moves 100 G samples/s with ONE lane per (b,q,head) unit):
//   k_da_cross_attn_bwd_unit  (A) the UNIT-OWNED gradients -- attention weights, sampling offsets, depth distribution.
//       They need the sampled VALUES (value loads, like the forward) but no scatter: the forward's lane mapping, its
//       chunk-major 16-byte corner loads, its XCD order; a lane owns its unit for every camera, so the per-sample
//       results are plain read-modify-writes in camera order (deterministic), batched per level through the lane's own
//       LDS row; the depth-distribution corners keep their fp32 global atomics.
//   k_da_cross_attn_bwd_scatter (B) the VALUE gradient -- needs NO value loads, only coordinates and weights: a lane owns a
//       unit (64 compacted hit queries of one head per wave), the head's gradient plane of the launch's token REGION
//       lives in LDS as 64-bit fixed point (see above), 4 corners x Dh ds_add_u64 per sample; per-region launches
//       cover pyramids whose whole plane does not fit (a region = whole levels, or a band of rows of one large level;
//       corners outside the launch's token range are left to the launch that owns them).
template <int DH>
__device__ __forceinline__ void fbbev_unit_sample_grad(const float* __restrict__ value, unsigned lane_off, const fbbev_bilinear& s,
                                                       int chunk_stride, const float (&g)[DH], float& dot, float& gx, float& gy) {
    constexpr int DHP = (DH + 3) / 4 * 4;
    const char* vb = reinterpret_cast<const char*>(value);
    const bool k1 = s.o1 >= 0, k2 = s.o2 >= 0, k3 = s.o3 >= 0, k4 = s.o4 >= 0;
    const unsigned b1 = lane_off + (k1 ? (unsigned)s.o1 * 4u : 0u), b2 = lane_off + (k2 ? (unsigned)s.o2 * 4u : 0u);
    const unsigned b3 = lane_off + (k3 ? (unsigned)s.o3 * 4u : 0u), b4 = lane_off + (k4 ? (unsigned)s.o4 * 4u : 0u);
    const unsigned cs = (unsigned)chunk_stride * 4u;
    fbbev_v4f a1[DHP / 4], a2[DHP / 4], a3[DHP / 4], a4[DHP / 4];
#pragma unroll
    for (int k = 0; k < DHP / 4; ++k) {
        a1[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b1 + k * cs));
        a2[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b2 + k * cs));
        a3[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b3 + k * cs));
        a4[k] = *reinterpret_cast<const fbbev_v4f*>(vb + (b4 + k * cs));
    }
    dot = gx = gy = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        const float v1 = k1 ? a1[c >> 2][c & 3] : 0.f, v2 = k2 ? a2[c >> 2][c & 3] : 0.f;
        const float v3 = k3 ? a3[c >> 2][c & 3] : 0.f, v4 = k4 ? a4[c >> 2][c & 3] : 0.f;
        dot += g[c] * (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4);
        gy += g[c] * (-s.hw * v1 - s.lw * v2 + s.hw * v3 + s.lw * v4);
        gx += g[c] * (-s.hh * v1 + s.hh * v2 - s.lh * v3 + s.lh * v4);
    }
}

template <int DH, bool QI>
__global__ void __launch_bounds__(256)
k_da_cross_attn_bwd_unit(long long n_units, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                         const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                         const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                         const float* __restrict__ qdepth, const float* __restrict__ offsets,
                         const float* __restrict__ attn, const float* __restrict__ grad_slots, int B, int Ncam, int S,
                         int M, int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor, int HS,
                         float* __restrict__ grad_pred_depth, float* __restrict__ grad_offsets,
                         float* __restrict__ grad_attn, unsigned int* __restrict__ gmax_bits) {
    // gmax_bits (may be null): max |grad_slots| of the call, folded here because this kernel reads every gradient row anyway --
    // the fixed-point scale of k_da_bwd_scatter_owned (bits of a non-negative float order like the float; non-finite -> inf)
    float gm_lane = 0.f;
    bool fin_lane = true;
    int gm_b = -1;                                             // the sample the running maximum belongs to (round 5: one scale per sample)
    auto flush_gmax = [&](int bb, float gm, bool fin) {
        if (!fin) gm = __builtin_inff();
        unsigned int gb;
        __builtin_memcpy(&gb, &gm, 4);
        if (gb != 0u) atomicMax(gmax_bits + bb, gb);
    };
    const int head_off_m = QI ? 4 : HS, chunk_stride = QI ? M * 4 : 4;
    const int row_stride = M * HS;
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int LP = L * P, LW = 4 * P + 1;
    float* row = fbbev_dyn_lds_f32() + threadIdx.x * LW;       // this lane's own row: [P] weights of the level, [3P] results
    const long long n_wg = (n_units + blockDim.x - 1) / blockDim.x, per_xcd = (n_wg + 7) / 8;
    for (long long w = blockIdx.x; (w >> 3) < per_xcd; w += gridDim.x) {
        const long long unit = ((w & 7) * per_xcd + (w >> 3)) * blockDim.x + threadIdx.x;
        if (unit >= n_units) continue;
        const int m = (int)(unit % M);
        const long long bq = unit / M;
        const int q = (int)(bq % Q);
        const int b = (int)(bq / Q);
        if (gmax_bits) {
            if (b != gm_b) {                                   // a lane's units come in ascending order: a new sample at most B times
                if (gm_b >= 0) flush_gmax(gm_b, gm_lane, fin_lane);
                gm_b = b; gm_lane = 0.f; fin_lane = true;
            }
            const float* gs = grad_slots + unit * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 2) {
                const fbbev_v2f t = *reinterpret_cast<const fbbev_v2f*>(gs + c);
                const float a0 = fabsf(t[0]), a1 = fabsf(t[1]);
                fin_lane = fin_lane && (a0 < __builtin_inff()) && (a1 < __builtin_inff());
                gm_lane = fmaxf(gm_lane, fmaxf(a0, a1));
            }
        }
        int count = 0;
        for (int cam = 0; cam < Ncam; ++cam) {
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            bool hit = false;
            for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            count += hit ? 1 : 0;
        }
        if (count == 0) continue;
        float g[DH];
        {
            const float inv = (float)count;
            const float* gs = grad_slots + unit * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 2) {
                const fbbev_v2f t = *reinterpret_cast<const fbbev_v2f*>(gs + c);
                g[c] = t[0] / inv; g[c + 1] = t[1] / inv;
            }
        }
        const long long wo0 = (head_minor & 1) ? bq * LP * M + m : unit * LP, wa0 = (head_minor & 2) ? bq * LP * M + m : unit * LP;
        const int wo_step = (head_minor & 1) ? M : 1, wa_step = (head_minor & 2) ? M : 1;
        const fbbev_v2f* op = reinterpret_cast<const fbbev_v2f*>(offsets) + wo0;
        for (int cam = 0; cam < Ncam; ++cam) {
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            bool hit = false;
            for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            if (!hit) continue;
            const long long bn = (long long)b * Ncam + cam;
            float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA], ddw[FBBEV_DA_MAX_ZA];
            int bin[FBBEV_DA_MAX_ZA];
            for (int z = 0; z < Za; ++z) {
                rx[z] = ref_cam[(base + z) * 2];
                ry[z] = ref_cam[(base + z) * 2 + 1];
                float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                bin[z] = (int)fb;
                dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + bin[z]) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
                ddw[z] = 0.f;
            }
            fbbev_v2f o_next = op[0];
            int lp = 0;
            for (int l = 0; l < L; ++l) {
                const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                const unsigned lane_off = (unsigned)(((bn * S + level_start[l]) * row_stride + m * head_off_m) * 4);
                for (int p = 0; p < P; ++p) row[p] = attn[wa0 + (long long)(lp + p) * wa_step];   // the level's weights
                for (int p = 0; p < P; ++p, ++lp) {
                    const fbbev_v2f o = o_next;
                    o_next = op[(long long)(lp + 1 < LP ? lp + 1 : lp) * wo_step];
                    const int z = p % Za;
                    const float loc_w = rx[z] + __fdiv_rn(o[0], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(o[1], (float)sh);
                    const float a = row[p];
                    const float weight = a * dw[z];
                    const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                    float ra = 0.f, rgx = 0.f, rgy = 0.f;
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw) {
                        const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, row_stride);
                        float dot, gx, gy;
                        fbbev_unit_sample_grad<DH>(value, lane_off, s, chunk_stride, g, dot, gx, gy);
                        ra = dw[z] * dot; rgx = weight * gx; rgy = weight * gy;
                        ddw[z] += a * dot;
                    }
                    row[P + 3 * p] = ra; row[P + 3 * p + 1] = rgx; row[P + 3 * p + 2] = rgy;
                }
                // one add per camera that sees the query, in camera order (the lane is the unit's only writer): all loads
                // of the level first, then the adds (a zero is not added: the sample was outside the image)
                {
                    // straight-line code (no branch between the loads and the stores: with conditional stores the compiler
                    // waited for EVERY earlier store before each next one -- 16 serialised round trips per level): slots
                    // beyond P repeat slot P-1 (same address, same old value, same sum: an idempotent second store), and
                    // a sample outside the image adds 0 (x + 0 == x)
                    float ca[FBBEV_DA_BWD_MAXP];
                    fbbev_v2f co[FBBEV_DA_BWD_MAXP];
                    long long ia[FBBEV_DA_BWD_MAXP], io[FBBEV_DA_BWD_MAXP];
#pragma unroll
                    for (int p = 0; p < FBBEV_DA_BWD_MAXP; ++p) {
                        const int pp = p < P ? p : P - 1;
                        ia[p] = wa0 + (long long)(lp - P + pp) * wa_step;
                        io[p] = (wo0 + (long long)(lp - P + pp) * wo_step) * 2;
                        ca[p] = grad_attn[ia[p]];
                        co[p] = *reinterpret_cast<const fbbev_v2f*>(grad_offsets + io[p]);
                    }
#pragma unroll
                    for (int p = 0; p < FBBEV_DA_BWD_MAXP; ++p) {
                        const int pp = p < P ? p : P - 1;
                        ca[p] += row[P + 3 * pp];
                        co[p][0] += row[P + 3 * pp + 1];
                        co[p][1] += row[P + 3 * pp + 2];
                    }
#pragma unroll
                    for (int p = 0; p < FBBEV_DA_BWD_MAXP; ++p) {
                        grad_attn[ia[p]] = ca[p];
                        *reinterpret_cast<fbbev_v2f*>(grad_offsets + io[p]) = co[p];
                    }
                }
            }
            // dw[z] -> the four corners of the query's bin plane (fbbev_plane_sample).  The M heads of a query are M adjacent
            // lanes with the SAME camera set, anchors, bins and corners: their ddw are summed across the lanes first and one
            // lane issues the atomics -- M x fewer of them (the scattered fp32 global atomics were 0.36 of this kernel's
            // 0.50 ms at the shipped shape: 8.7 M of them)
            const bool reduce_heads = M <= 64 && (M & (M - 1)) == 0 && (64 % M) == 0;
            for (int z = 0; z < Za; ++z) {
                float dsum = ddw[z];
                if (reduce_heads)
                    for (int o = 1; o < M; o <<= 1) dsum += __shfl_xor(dsum, o, 64);
                if (reduce_heads && m != 0) continue;
                const float h_im = ry[z] * H0 - 0.5f, w_im = rx[z] * W0 - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H0 && w_im < (float)W0) || dsum == 0.f) continue;
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, H0, W0, 1);
                float* gd = grad_pred_depth + (bn * DC + bin[z]) * (long long)(H0 * W0);
                if (s.o1 >= 0) fbbev_atomic_add_f32(gd + s.o1, s.w1 * dsum);
                if (s.o2 >= 0) fbbev_atomic_add_f32(gd + s.o2, s.w2 * dsum);
                if (s.o3 >= 0) fbbev_atomic_add_f32(gd + s.o3, s.w3 * dsum);
                if (s.o4 >= 0) fbbev_atomic_add_f32(gd + s.o4, s.w4 * dsum);
            }
        }
    }
    if (gmax_bits) {                                                    // every lane is back here
        // one atomic per wave when its lanes ended in the same sample (the rule; a wave that straddles two samples: one per lane)
        int lo = gm_b < 0 ? 0x7fffffff : gm_b, hi = gm_b;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
            lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
        }
        if (hi >= 0 && lo == hi) {
            if (!fin_lane) gm_lane = __builtin_inff();
            if (gm_b < 0) gm_lane = 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gm_lane = fmaxf(gm_lane, __shfl_xor(gm_lane, o, 64));
            if ((threadIdx.x & 63) == 0) flush_gmax(hi, gm_lane, true);
        } else if (gm_b >= 0) {
            flush_gmax(gm_b, gm_lane, fin_lane);
        }
    }
}

// Pre-pass of the scatter (several region launches; FBBEV_DA_BWD_PREPASS=0 turns it off): what every region launch of k_da_cross_attn_bwd_scatter recomputes per
// (camera, query) and that does not depend on the region -- does the camera see the query, how many cameras do (the reference
// divides the slot by that count), the depth weight of each of the Za anchors (a bilinear sample of the predicted depth
// distribution) -- computed ONCE: info[((b*Ncam + cam)*Q + q)*IS + {0: count as float (0 = not seen), 1 + z: dw[z]}].
// One lane per (b, q), looping over the cameras.  The per-launch times of the 6-region configs[2] pyramid (428 / 4 x 305 / 732 us)
// say ~260 us of each launch is region-independent work; this part of it is worth 0.2 ms of the 2.43 ms (measured).
__global__ void __launch_bounds__(256)
k_da_bwd_hitinfo(const int64_t* __restrict__ spatial_shapes, const float* __restrict__ pred_depth,
                 const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask, const float* __restrict__ qdepth,
                 int B, int Ncam, int Q, int Za, int DC, float d0, float dstep, int IS, float* __restrict__ info) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Q) return;
    const int q = (int)(i % Q), b = (int)(i / Q);
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    int count = 0;
    for (int c2 = 0; c2 < Ncam; ++c2) {
        const long long b2 = (((long long)c2 * B + b) * Q + q) * Za;
        bool h2 = false;
        for (int z = 0; z < Za; ++z) h2 = h2 || (mask[b2 + z] != 0);
        count += h2 ? 1 : 0;
    }
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long base = (((long long)cam * B + b) * Q + q) * Za;
        const long long bn = (long long)b * Ncam + cam;
        float* dst = info + (bn * Q + q) * IS;
        bool hit = false;
        for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
        dst[0] = hit ? (float)(count > 1 ? count : 1) : 0.f;
        if (!hit) continue;
        for (int z = 0; z < Za; ++z) {
            const float rx = ref_cam[(base + z) * 2], ry = ref_cam[(base + z) * 2 + 1];
            float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
            fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
            dst[1 + z] = fbbev_plane_sample(pred_depth + (bn * DC + (int)fb) * (long long)(H0 * W0), H0, W0, rx, ry);
        }
    }
}

// (B) value-gradient scatter of one token REGION [tok0, tok1) = levels [lvl0, lvl1) (a band of rows when one level is split
// over several launches).  Workgroup = (sample b, head m, chunk of consecutive BEV queries); a lane = one query the camera
// sees (compacted per camera); plane = (tok1 - tok0) x HS 64-bit fixed-point words in LDS; part[b][m][chunk][cam][S*HS] gets
// the region's slice.  No value loads, no shuffles, no global atomics.
template <int NT, int DH>
__global__ void __launch_bounds__(NT)
k_da_cross_attn_bwd_scatter(const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
                            const float* __restrict__ pred_depth, const float* __restrict__ ref_cam,
                            const unsigned char* __restrict__ mask, const float* __restrict__ qdepth,
                            const float* __restrict__ offsets, const float* __restrict__ attn,
                            const float* __restrict__ grad_slots, int B, int Ncam, int S, int M, int L, int Q, int P, int Za,
                            int DC, float d0, float dstep, int head_minor, int HS, int n_chunks, int q_per_chunk, int lvl0,
                            int lvl1, int tok0, int tok1, int copies, float* __restrict__ part, const float* __restrict__ info,
                            int IS) {
    // [tok1 - tok0][HS] fixed point, skewed by one word every 8 tokens: a token pitch of HS 64-bit words (24 banks at HS = 12)
    // repeats its bank every 8 tokens; the skew breaks the period (SQ_LDS_BANK_CONFLICT was 2x the busy cycles) for 1 % more LDS
    // `copies` planes when the region is small (the coarse levels of a pyramid: a few hundred tokens that EVERY query of the
    // chunk samples): lane l adds into copy l % copies, the flush sums the copies -- same-address lanes of one ds_add_u64
    // serialise (16 lanes per address: 1.2 instead of 22 lane-adds per ns)
    long long* plane0 = reinterpret_cast<long long*>(fbbev_dyn_lds_f32());
    const int plane_n = (tok1 - tok0) * HS, plane_w = FBBEV_DA_PLANE_WORDS(tok1 - tok0, HS);
    long long* plane = plane0 + (threadIdx.x % copies) * plane_w;
    unsigned short* hits = reinterpret_cast<unsigned short*>(plane0 + copies * plane_w);   // [q_per_chunk] chunk-relative queries the camera sees
    int* n_hits = reinterpret_cast<int*>(hits + ((q_per_chunk + 1) & ~1));       // [1]
    float* red = reinterpret_cast<float*>(n_hits + 1);                           // [NT/64] block maximum
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x % n_chunks;
    const int m = (blockIdx.x / n_chunks) % M;
    const int b = blockIdx.x / (n_chunks * M);
    const int q0 = chunk * q_per_chunk, q1 = (q0 + q_per_chunk < Q) ? q0 + q_per_chunk : Q;
    const int nq = q1 - q0;
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int LP = L * P;
    for (int i = threadIdx.x; i < copies * plane_w; i += NT) plane0[i] = 0ll;
    // scale of the fixed-point plane: sc = 2^(30 - ex) with max|grad_slots| < 2^ex over the chunk's units (see above)
    float gmax = 0.f;
    bool finite = true;
    for (int i = threadIdx.x; i < nq * DH; i += NT) {
        const int qi = i / DH, c = i - qi * DH;
        const float v = fabsf(grad_slots[(((long long)b * Q + q0 + qi) * M + m) * DH + c]);
        finite = finite && (v < __builtin_inff());
        gmax = fmaxf(gmax, v);
    }
    if (!finite) gmax = __builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
    if (lane == 0) red[threadIdx.x >> 6] = gmax;
    __syncthreads();
    gmax = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) gmax = fmaxf(gmax, red[w]);
    const bool poisoned = !(gmax < __builtin_inff());
    float sc = 0.f, inv_sc = 0.f;
    if (!poisoned && gmax > 0.f) {
        unsigned int gb;
        __builtin_memcpy(&gb, &gmax, 4);
        int ex = (int)((gb >> 23) & 255u) - 126;
        if (ex < -90) ex = -90;
        const unsigned int sb = (unsigned int)(127 + 30 - ex) << 23, ib = (unsigned int)(127 - 30 + ex) << 23;
        __builtin_memcpy(&sc, &sb, 4);
        __builtin_memcpy(&inv_sc, &ib, 4);
    }
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long bn = (long long)b * Ncam + cam;
        if (threadIdx.x == 0) *n_hits = 0;
        __syncthreads();
        for (int i0 = 0; i0 < nq; i0 += NT) {
            const int qi = i0 + threadIdx.x;
            bool hit = false;
            if (qi < nq) {
                if (info) {                                                  // pre-pass: one float says whether the camera sees the query
                    hit = info[(bn * Q + (q0 + qi)) * IS] != 0.f;
                } else {
                    const long long base = (((long long)cam * B + b) * Q + (q0 + qi)) * Za;
                    for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
                }
            }
            const unsigned long long bal = __ballot(hit ? 1 : 0);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(n_hits, __popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (hit) hits[wbase + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = (unsigned short)qi;
        }
        __syncthreads();
        const int nh = *n_hits;
        for (int it0 = 0; it0 < nh; it0 += NT) {
            // lane -> hit: a stride of 37 (41 when 37 divides the block) inside each block of NT hits.  Neighbouring BEV
            // queries sample the same camera tokens: with consecutive hits on consecutive lanes up to 16 lanes of one
            // ds_add_u64 share an ADDRESS and serialise (1.2 instead of 22 lane-adds per ns, profiles/r02_micro_lds_atomics.jsonl)
            const int nblk = nh - it0 < NT ? nh - it0 : NT;
            if ((int)threadIdx.x >= nblk) continue;
            const int it = it0 + (int)(((unsigned)threadIdx.x * (nblk % 37 == 0 ? 41u : 37u)) % (unsigned)nblk);
            const int q = q0 + (int)hits[it];
            const long long bq = (long long)b * Q + q;
            const long long u = bq * M + m;
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            float inv;
            float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA];
            if (info) {
                const float* ip = info + (bn * Q + q) * IS;
                inv = ip[0];
                for (int z = 0; z < Za; ++z) {
                    rx[z] = ref_cam[(base + z) * 2];
                    ry[z] = ref_cam[(base + z) * 2 + 1];
                    dw[z] = ip[1 + z];
                }
            } else {
                int count = 0;
                for (int c2 = 0; c2 < Ncam; ++c2) {
                    const long long b2 = (((long long)c2 * B + b) * Q + q) * Za;
                    bool h2 = false;
                    for (int z = 0; z < Za; ++z) h2 = h2 || (mask[b2 + z] != 0);
                    count += h2 ? 1 : 0;
                }
                inv = (float)(count > 1 ? count : 1);
                for (int z = 0; z < Za; ++z) {
                    rx[z] = ref_cam[(base + z) * 2];
                    ry[z] = ref_cam[(base + z) * 2 + 1];
                    float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                    fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                    dw[z] = fbbev_plane_sample(pred_depth + (bn * DC + (int)fb) * (long long)(H0 * W0), H0, W0, rx[z], ry[z]);
                }
            }
            float gs[DH];
#pragma unroll
            for (int c = 0; c < DH; ++c) gs[c] = grad_slots[u * DH + c] / inv * sc;          // sc is a power of two: exact
            const long long wo0 = (head_minor & 1) ? bq * LP * M + m : u * LP, wa0 = (head_minor & 2) ? bq * LP * M + m : u * LP;
            const int wo_step = (head_minor & 1) ? M : 1, wa_step = (head_minor & 2) ? M : 1;
            const int lp0 = lvl0 * P, lp1 = lvl1 * P;
            fbbev_v2f o_next = *reinterpret_cast<const fbbev_v2f*>(offsets + (wo0 + (long long)lp0 * wo_step) * 2);
            float a_next = attn[wa0 + (long long)lp0 * wa_step];
            int lp = lp0;
            for (int l = lvl0; l < lvl1; ++l) {
                const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                const int ls = (int)level_start[l];
                for (int p = 0; p < P; ++p, ++lp) {
                    const int nlp = lp + 1 < lp1 ? lp + 1 : lp;
                    const fbbev_v2f o = o_next;
                    const float a = a_next;
                    o_next = *reinterpret_cast<const fbbev_v2f*>(offsets + (wo0 + (long long)nlp * wo_step) * 2);
                    a_next = attn[wa0 + (long long)nlp * wa_step];
                    const int z = p % Za;
                    const float loc_w = rx[z] + __fdiv_rn(o[0], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(o[1], (float)sh);
                    const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw)) continue;
                    const float weight = a * dw[z];
                    const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, 1);          // o1..o4 = token indices of the level
                    const int t1 = ls + s.o1 - tok0, t2 = ls + s.o2 - tok0, t3 = ls + s.o3 - tok0, t4 = ls + s.o4 - tok0;
                    const int span = tok1 - tok0;
                    const bool k1 = s.o1 >= 0 && t1 >= 0 && t1 < span, k2 = s.o2 >= 0 && t2 >= 0 && t2 < span;
                    const bool k3 = s.o3 >= 0 && t3 >= 0 && t3 < span, k4 = s.o4 >= 0 && t4 >= 0 && t4 < span;
                    // corner outer, channel inner: ONE divergent region per corner instead of one per (corner, channel) -- the
                    // 40 predicated adds of a sample were 40 exec-mask diamonds (~600 instructions per sample, 33 M scalar
                    // instructions per launch, profiles/r04_diag_da_bwd_regions.json).  Same product order as the other
                    // kernels: w_corner * (g * weight); integer adds commute, so the plane is bit-identical.
                    float tg[DH];
#pragma unroll
                    for (int c = 0; c < DH; ++c) tg[c] = gs[c] * weight;
                    fbbev_lds_corner_add<DH>(k1, plane + FBBEV_DA_PLANE_IDX(t1, HS), s.w1, tg);
                    fbbev_lds_corner_add<DH>(k2, plane + FBBEV_DA_PLANE_IDX(t2, HS), s.w2, tg);
                    fbbev_lds_corner_add<DH>(k3, plane + FBBEV_DA_PLANE_IDX(t3, HS), s.w3, tg);
                    fbbev_lds_corner_add<DH>(k4, plane + FBBEV_DA_PLANE_IDX(t4, HS), s.w4, tg);
                }
            }
        }
        __syncthreads();
        float* dst = part + ((((long long)b * M + m) * n_chunks + chunk) * Ncam + cam) * (long long)S * HS + (long long)tok0 * HS;
        for (int i = threadIdx.x * 4; i < plane_n; i += NT * 4) {            // HS % 4 == 0: a group of 4 stays inside one token
            long long* src = plane0 + FBBEV_DA_PLANE_IDX(i / HS, HS) + (i % HS);
            fbbev_v4f t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                long long acc = 0;
                for (int cp = 0; cp < copies; ++cp) { acc += src[cp * plane_w + e]; src[cp * plane_w + e] = 0ll; }
                t[e] = poisoned ? __builtin_nanf("") : (float)acc * inv_sc;               // one rounding (int64 -> fp32)
            }
            *reinterpret_cast<fbbev_v4f*>(dst + i) = t;
        }
        __syncthreads();
    }
}

// grad_value[(b*Ncam+cam), s, (m,c) in the layout of value] = sum over the query chunks of part[b][m][chunk][cam][s*HS+c]
__global__ void __launch_bounds__(256)
k_da_bwd_reduce(const float* __restrict__ part, int B, int Ncam, int S, int M, int HS, int n_chunks, int interleaved,
                float* __restrict__ grad_value) {
    const long long n = (long long)B * Ncam * S * M * HS;
    const long long plane_n = (long long)S * HS;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates (b, cam, m, s, c) with c fastest: reads of consecutive lanes are consecutive floats of a plane
        const int c = (int)(idx % HS);
        long long r = idx / HS;
        const int sidx = (int)(r % S); r /= S;
        const int m = (int)(r % M); r /= M;
        const int cam = (int)(r % Ncam);
        const int b = (int)(r / Ncam);
        const float* src = part + (((long long)b * M + m) * n_chunks * Ncam + cam) * plane_n + (long long)sidx * HS + c;
        float acc = 0.f;
        for (int k = 0; k < n_chunks; ++k) acc += src[(long long)k * Ncam * plane_n];
        const long long row = (((long long)b * Ncam + cam) * S + sidx) * (long long)(M * HS);
        grad_value[row + (interleaved ? (c >> 2) * (M * 4) + m * 4 + (c & 3) : m * HS + c)] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: value-gradient scatter with OUTPUT-OWNED planes.  The chunked scatter above gives a workgroup a chunk of queries
// and walks the cameras (a barrier-bracketed compaction + a mostly empty tail iteration + a 34 KB flush per camera, one launch
// per token region, 551 MB of per-chunk partial planes and a reduction at the configs[2] pyramid: 2.4 ms, waves parked 63 %,
// profiles/r04_diag_da_bwd_regions.json -- halving its instruction count changed nothing).  Here a workgroup OWNS the plane of
// one (sample, camera, head, token region) and walks the camera's whole hit list: one long uniform loop, no per-camera
// barriers, no partial planes, no reduction (integer adds commute, so any list order gives the same bits), every region of the
// pyramid in ONE launch, and small regions (coarse levels) keep up to 16 copies of their plane against same-address adds.
//   k_da_bwd_init      zero the per-(sample, camera) hit counters and the |gradient| maximum (k_da_cross_attn_bwd_unit folds
//                      max |grad_slots| of the call into it: the planes' fixed-point scale)
//   k_da_bwd_hitlist   per (sample, query): camera count / depth weights (the table of k_da_bwd_hitinfo), append the query to the
//                      list of every camera that sees it
//   k_da_bwd_scatter_owned
#define FBBEV_DA_HIT_ZA 4                                   // anchors a hit record holds
#define FBBEV_DA_HIT_REC (2 + 3 * FBBEV_DA_HIT_ZA + 2)        // floats per record (64 bytes): q, cameras, dw[4], rx[4], ry[4], pad
struct fbbev_da_bwd_region_tab { int n; unsigned int perm; int lvl0[24], lvl1[24], tok0[24], tok1[24], copies[24]; };

__global__ void __launch_bounds__(256)
k_da_bwd_init(int n_count, int* __restrict__ hit_count, int n_max, unsigned int* __restrict__ gmax_bits) {
    for (int i = threadIdx.x; i < n_count; i += 256) hit_count[i] = 0;
    for (int i = threadIdx.x; i < n_max; i += 256) gmax_bits[i] = 0u;
}

// grid B * ceil(Q / 256): a wave never straddles two samples, so one ballot + one counter add per (wave, camera)
__global__ void __launch_bounds__(256)
k_da_bwd_hitlist(const int64_t* __restrict__ spatial_shapes, const float* __restrict__ pred_depth,
                 const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask, const float* __restrict__ qdepth,
                 int B, int Ncam, int Q, int Za, int DC, float d0, float dstep, float* __restrict__ hit_rec,
                 int* __restrict__ hit_count) {
    const int bps = (Q + 255) / 256;                                      // blocks per sample
    const int b = blockIdx.x / bps, lane = threadIdx.x & 63;
    const int qb = (blockIdx.x - b * bps) * 256, q = qb + (int)threadIdx.x;
    const bool live = q < Q;
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    int count = 0;
    if (live) {
        for (int c2 = 0; c2 < Ncam; ++c2) {
            const long long b2 = (((long long)c2 * B + b) * Q + q) * Za;
            bool h2 = false;
            for (int z = 0; z < Za; ++z) h2 = h2 || (mask[b2 + z] != 0);
            count += h2 ? 1 : 0;
        }
    }
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long bn = (long long)b * Ncam + cam;
        bool hit = false;
        float rec[FBBEV_DA_HIT_REC];
        if (live) {
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            if (hit) {
                __builtin_memcpy(&rec[0], &q, 4);
                rec[1] = (float)(count > 1 ? count : 1);
                for (int k = 2; k < FBBEV_DA_HIT_REC; ++k) rec[k] = 0.f;        // anchors beyond Za and the two pad words
                for (int z = 0; z < Za; ++z) {
                    const float rx = ref_cam[(base + z) * 2], ry = ref_cam[(base + z) * 2 + 1];
                    float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                    fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                    rec[2 + z] = fbbev_plane_sample(pred_depth + (bn * DC + (int)fb) * (long long)(H0 * W0), H0, W0, rx, ry);
                    rec[2 + FBBEV_DA_HIT_ZA + z] = rx;
                    rec[2 + 2 * FBBEV_DA_HIT_ZA + z] = ry;
                }
            }
        }
        const unsigned long long bal = __ballot(hit ? 1 : 0);
        int wbase = 0;
        if (lane == 0 && bal) wbase = atomicAdd(hit_count + bn, __popcll(bal));
        wbase = __shfl(wbase, 0, 64);
        if (hit) {
            // the hit's record in LIST order: {query, cameras that see it, depth weight / reference x / y of the anchors} -- the scatter
            // reads its hits as consecutive 64-byte records instead of chasing list -> query -> table / reference rows
            float* dst = hit_rec + (bn * Q + wbase + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))) * FBBEV_DA_HIT_REC;
#pragma unroll
            for (int k = 0; k < FBBEV_DA_HIT_REC; k += 4) *reinterpret_cast<fbbev_v4f*>(dst + k) = fbbev_v4f{rec[k], rec[k + 1], rec[k + 2], rec[k + 3]};
        }
    }
}

// One workgroup = (token region r, sample b, camera cam, head m).  blockIdx -> (r, b, cam, m): the workgroups of one
// (sample, camera) -- which share the hit list, the table, the reference points and the gradient rows -- sit on one XCD when the
// number of (sample, camera) pairs divides by 8.  grad_value gets the region's tokens of head m directly.
template <int NT, int DH>
__global__ void __launch_bounds__(NT)
k_da_bwd_scatter_owned(const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
                       const float* __restrict__ ref_cam, const float* __restrict__ offsets, const float* __restrict__ attn,
                       const float* __restrict__ grad_slots, int B, int Ncam, int S, int M, int L, int Q, int P, int Za,
                       int head_minor, int HS, fbbev_da_bwd_region_tab tab, const float* __restrict__ hit_rec,
                       const int* __restrict__ hit_count,
                       const unsigned int* __restrict__ gmax_bits, int interleaved, float* __restrict__ grad_value) {
    const int pairs = B * Ncam;
    int r, pair, m;
    if (pairs % 8 == 0) {
        const int ppx = pairs / 8, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pair = xcd * ppx + slot % ppx;
        const int rest = slot / ppx;
        m = rest % M;
        r = rest / M;
    } else {
        pair = blockIdx.x % pairs;
        const int rest = blockIdx.x / pairs;
        m = rest % M;
        r = rest / M;
    }
    const int b = pair / Ncam, cam = pair - b * Ncam;
    const int lvl0 = tab.lvl0[r], lvl1 = tab.lvl1[r], tok0 = tab.tok0[r], tok1 = tab.tok1[r], copies = tab.copies[r];
    long long* plane0 = reinterpret_cast<long long*>(fbbev_dyn_lds_f32());
    const int span = tok1 - tok0, plane_n = span * HS, plane_w = FBBEV_DA_PLANE_WORDS(span, HS);
    long long* plane = plane0 + (threadIdx.x % copies) * plane_w;
    for (int i = threadIdx.x; i < copies * plane_w; i += NT) plane0[i] = 0ll;
    // scale of the fixed-point plane: sc = 2^(30 - ex) with max |grad_slots| of the SAMPLE < 2^ex (every addend is
    // |corner weight * attention * depth weight / cameras| <= 1 times a gradient: below 2^30; a token takes < 2^33 of them)
    const unsigned int gb = gmax_bits[b];                      // round 5: the scale of THIS sample's planes (one outlier
    const bool poisoned = gb >= 0x7f800000u;                   // gradient no longer sets the quantum of every other sample)
    float sc = 0.f, inv_sc = 0.f;
    if (!poisoned && gb != 0u) {
        int ex = (int)((gb >> 23) & 255u) - 126;
        if (ex < -90) ex = -90;
        const unsigned int sb = (unsigned int)(127 + 30 - ex) << 23, ib = (unsigned int)(127 - 30 + ex) << 23;
        __builtin_memcpy(&sc, &sb, 4);
        __builtin_memcpy(&inv_sc, &ib, 4);
    }
    const long long bn = (long long)b * Ncam + cam;
    const int nh = (poisoned || sc == 0.f) ? 0 : hit_count[bn];
    const float* list = hit_rec + bn * (long long)Q * FBBEV_DA_HIT_REC;
    const int LP = L * P;
    const bool perm = (tab.perm >> r) & 1;
    __syncthreads();
    // Round 6: the hit walk was a chain of ~2 + P dependent round trips per hit (record -> gradient row -> one point's offset / weight
    // at a time, each behind the previous point's adds): "a pass over the hit lists costs ~200 us whatever fraction of its corners is
    // live" (round 4) was this latency, not the LDS adds.  Now the NEXT block's record is requested before the current hit is worked
    // on, and a hit's gradient row and the offsets / weights of up to eight points are requested together: two round trips per hit,
    // one of them hidden.  Same expressions, same order of adds per lane.
    constexpr int PB = 8;
    auto hit_index = [&](int it0) {
        // lane -> hit: a stride of 37 (41 when 37 divides the block) inside each block of NT hits -- neighbouring queries sample
        // the same tokens, consecutive hits on consecutive lanes would share addresses inside one ds_add_u64
        const int nblk = nh - it0 < NT ? nh - it0 : NT;
        if ((int)threadIdx.x >= nblk) return -1;
        return it0 + (perm ? (int)(((unsigned)threadIdx.x * (nblk % 37 == 0 ? 41u : 37u)) % (unsigned)nblk) : (int)threadIdx.x);
    };
    fbbev_v4f n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0, n2 = n0, n3 = n0;
    {
        const int it = nh > 0 ? hit_index(0) : -1;
        const float* rp = list + (long long)(it < 0 ? 0 : it) * FBBEV_DA_HIT_REC;       // 64 bytes, 16-byte aligned
        if (nh > 0) {
            n0 = *reinterpret_cast<const fbbev_v4f*>(rp); n1 = *reinterpret_cast<const fbbev_v4f*>(rp + 4);
            n2 = *reinterpret_cast<const fbbev_v4f*>(rp + 8); n3 = *reinterpret_cast<const fbbev_v4f*>(rp + 12);
        }
    }
    for (int it0 = 0; it0 < nh; it0 += NT) {
        const bool mine = hit_index(it0) >= 0;
        const fbbev_v4f r0 = n0, r1 = n1, r2 = n2, r3 = n3;
        if (it0 + NT < nh) {                                                 // uniform: the next block's record of this lane
            const int itn = hit_index(it0 + NT);
            const float* rp = list + (long long)(itn < 0 ? 0 : itn) * FBBEV_DA_HIT_REC;
            n0 = *reinterpret_cast<const fbbev_v4f*>(rp); n1 = *reinterpret_cast<const fbbev_v4f*>(rp + 4);
            n2 = *reinterpret_cast<const fbbev_v4f*>(rp + 8); n3 = *reinterpret_cast<const fbbev_v4f*>(rp + 12);
        }
        if (!mine) continue;
        const float q_bits = r0[0];
        int q;
        __builtin_memcpy(&q, &q_bits, 4);
        const float inv = r0[1];
        const long long bq = (long long)b * Q + q;
        const long long u = bq * M + m;
        float rx[FBBEV_DA_MAX_ZA], ry[FBBEV_DA_MAX_ZA], dw[FBBEV_DA_MAX_ZA];
        {
            const float rec[16] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3], r3[0], r3[1], r3[2], r3[3]};
#pragma unroll
            for (int z = 0; z < FBBEV_DA_MAX_ZA; ++z) {
                dw[z] = z < FBBEV_DA_HIT_ZA ? rec[2 + z] : 0.f;
                rx[z] = z < FBBEV_DA_HIT_ZA ? rec[2 + FBBEV_DA_HIT_ZA + z] : 0.f;
                ry[z] = z < FBBEV_DA_HIT_ZA ? rec[2 + 2 * FBBEV_DA_HIT_ZA + z] : 0.f;
            }
        }
        const long long wo0 = (head_minor & 1) ? bq * LP * M + m : u * LP, wa0 = (head_minor & 2) ? bq * LP * M + m : u * LP;
        const int wo_step = (head_minor & 1) ? M : 1, wa_step = (head_minor & 2) ? M : 1;
        float gs[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) gs[c] = grad_slots[u * DH + c];                     // (raw: scaled below, behind the batch's requests)
        bool scaled = false;
        for (int l = lvl0; l < lvl1; ++l) {
            const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
            const int ls = (int)level_start[l];
            for (int p0 = 0; p0 < P; p0 += PB) {
                fbbev_v2f ob[PB];
                float ab[PB];
#pragma unroll
                for (int e = 0; e < PB; ++e) {
                    const long long lp = (long long)l * P + (p0 + e < P ? p0 + e : p0);       // (clamped: unconditional loads)
                    ob[e] = *reinterpret_cast<const fbbev_v2f*>(offsets + (wo0 + lp * wo_step) * 2);
                    ab[e] = attn[wa0 + lp * wa_step];
                }
                if (!scaled) {
#pragma unroll
                    for (int c = 0; c < DH; ++c) gs[c] = gs[c] / inv * sc;                // sc is a power of two: exact
                    scaled = true;
                }
#pragma unroll
                for (int e = 0; e < PB; ++e) {
                    const int p = p0 + e;
                    if (p >= P) break;
                    const fbbev_v2f o = ob[e];
                    const float a = ab[e];
                    const int z = p % Za;
                    const float loc_w = rx[z] + __fdiv_rn(o[0], (float)sw);
                    const float loc_h = ry[z] + __fdiv_rn(o[1], (float)sh);
                    const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw)) continue;
                    const float weight = a * dw[z];
                    const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, 1);          // o1..o4 = token indices of the level
                    const int t1 = ls + s.o1 - tok0, t2 = ls + s.o2 - tok0, t3 = ls + s.o3 - tok0, t4 = ls + s.o4 - tok0;
                    const bool k1 = s.o1 >= 0 && t1 >= 0 && t1 < span, k2 = s.o2 >= 0 && t2 >= 0 && t2 < span;
                    const bool k3 = s.o3 >= 0 && t3 >= 0 && t3 < span, k4 = s.o4 >= 0 && t4 >= 0 && t4 < span;
                    float tg[DH];
#pragma unroll
                    for (int c = 0; c < DH; ++c) tg[c] = gs[c] * weight;                // w_corner * (g * weight): the order of the other kernels
                    fbbev_lds_corner_add<DH>(k1, plane + FBBEV_DA_PLANE_IDX(t1, HS), s.w1, tg);
                    fbbev_lds_corner_add<DH>(k2, plane + FBBEV_DA_PLANE_IDX(t2, HS), s.w2, tg);
                    fbbev_lds_corner_add<DH>(k3, plane + FBBEV_DA_PLANE_IDX(t3, HS), s.w3, tg);
                    fbbev_lds_corner_add<DH>(k4, plane + FBBEV_DA_PLANE_IDX(t4, HS), s.w4, tg);
                }
            }
        }
    }
    __syncthreads();
    const int MHS = M * HS;
    float* dst = grad_value + (bn * S + tok0) * (long long)MHS;
    for (int i = threadIdx.x * 4; i < plane_n; i += NT * 4) {            // HS % 4 == 0: a group of 4 stays inside one token
        const int t = i / HS, c = i - t * HS;
        const long long* src = plane0 + FBBEV_DA_PLANE_IDX(t, HS) + c;
        fbbev_v4f v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            long long acc = 0;
            for (int cp = 0; cp < copies; ++cp) acc += src[cp * plane_w + e];
            v[e] = poisoned ? __builtin_nanf("") : (float)acc * inv_sc;                   // one rounding (int64 -> fp32)
        }
        *reinterpret_cast<fbbev_v4f*>(dst + (long long)t * MHS + (interleaved ? (c >> 2) * (M * 4) + m * 4 : m * HS + c)) = v;
    }
}
