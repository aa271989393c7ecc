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
                    const float a = stage_attn ? my_attn[lp] : attn[(head_minor & 2) ? (bq * LP + lp) * M + m : unit * LP + lp];
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


// ---------------------------------------------------------------- backward with an LDS-resident value-gradient plane
// k_da_cross_attn_bwd sends every corner of every sample to the value gradient with a global fp32 atomic: at the shipped
// shapes (Q = 10^4, one 16x44 level) ~110 adds land on each of the 1.6 M gradient floats, across all 8 XCDs -- 1.5 ms at
// B = 4, bound by the atomic rate.  Here a workgroup owns (sample b, head m, a chunk of consecutive BEV queries) and
// walks the cameras in order; for each camera the head's gradient plane (S tokens x HS channels) lives in LDS, the corner
// adds are LDS atomics, and the finished plane is written with plain 16-byte stores to this workgroup's slice of a
// partial buffer  part[b][m][chunk][cam][S*HS].  k_da_bwd_reduce then sums the chunks into grad_value in the layout of
// `value` -- no global atomic touches the value gradient.
// The plane is FIXED POINT: 64-bit integers in units of 2^-30 of the power of two above max|grad_slots| of the chunk.
// ds_add_f32 retires ~0.8 lanes per ns and CU on gfx950, ds_add_u64 22 (profiles/r02_micro_lds_atomics.jsonl: the fp32
// LDS atomic is 40x slower than the integer ones), and integer adds commute: the value gradient is bit-reproducible run
// to run, which neither the fp32-atomic kernels here nor mmcv's col2im are.  A contribution w*g*attn*dw is at most
// max|g| in magnitude, so it is rounded ONCE to a multiple of 2^-30 of that bound (finer than its own fp32 ulp for
// everything within 2^-6 of the largest contribution) and the <= q_per_chunk*L*P adds of a plane cannot overflow 63 bits;
// the plane is converted back with one rounding.  Non-finite upstream gradients turn the chunk's planes into NaN.
// grad_attn / grad_offsets: plain read-modify-writes of the unit's owner (the same workgroup handles a unit for every
// camera, phases separated by barriers -> fixed order); grad_pred_depth: fp32 global atomics as in k_da_cross_attn_bwd.
// Lane mapping: FOUR lanes own a (b,q,m) unit -- lane k its channels 4k..4k+3 (one 16-byte load per corner; with the
// chunk-major token rows the three chunk lanes of a head read three 16-byte pieces) and its depth anchor z = k (k+4) --
// so a wave works on 16 queries at once; the queries a camera sees (~28 % of a chunk) are first compacted into an LDS
// list, so every group of every wave is busy.  (One lane per channel and one query per 16 lanes, the layout of the
// atomic kernel, left 62 % of the lanes and a third of the groups active: 0.99 ms at the shipped shape, B = 4.)
template <int NT>        // threads per workgroup: NT/4 units per iteration share one plane
__global__ void __launch_bounds__(NT)
k_da_cross_attn_bwd_tile(const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                         const int64_t* __restrict__ level_start, const float* __restrict__ pred_depth,
                         const float* __restrict__ ref_cam, const unsigned char* __restrict__ mask,
                         const float* __restrict__ qdepth, const float* __restrict__ offsets,
                         const float* __restrict__ attn, const float* __restrict__ grad_slots, int B, int Ncam, int S,
                         int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor,
                         int HS, int n_chunks, int q_per_chunk, float* __restrict__ part,
                         float* __restrict__ grad_pred_depth, float* __restrict__ grad_offsets,
                         float* __restrict__ grad_attn) {
    long long* plane = reinterpret_cast<long long*>(fbbev_dyn_lds_f32());     // [S][HS] fixed point
    const int plane_n = S * HS;
    unsigned short* hits = reinterpret_cast<unsigned short*>(plane + plane_n);   // [q_per_chunk] queries (chunk-relative) the camera sees
    int* n_hits = reinterpret_cast<int*>(hits + ((q_per_chunk + 1) & ~1));       // [1]
    float* red = reinterpret_cast<float*>(n_hits + 1);         // [NT/64] block maximum
    float* stage = red + NT / 64;                              // [NT/4 groups][L*P][3]: a unit's weight / offset gradients of one camera
    const int lane = threadIdx.x & 63;
    const int k = threadIdx.x & 3;                             // channel chunk / anchor lane of the group
    const int gidx = threadIdx.x >> 2;                         // group in the workgroup: 0..NT/4-1
    const int gbase = lane & ~3;                               // first lane of the group in its wave
    const int chunk = blockIdx.x % n_chunks;
    const int m = (blockIdx.x / n_chunks) % M;
    const int b = blockIdx.x / (n_chunks * M);
    const int q0 = chunk * q_per_chunk, q1 = (q0 + q_per_chunk < Q) ? q0 + q_per_chunk : Q;
    const int nq = q1 - q0;
    const int row_stride = M * HS;
    const int H0 = (int)spatial_shapes[0], W0 = (int)spatial_shapes[1];
    const int lane_off = (head_minor & 4) ? k * (M * 4) + m * 4 : m * HS + 4 * k;   // this lane's 4 channels in a token row
    const bool chunk_live = 4 * k < Dh;                        // HS may hold a chunk of pure padding (Dh = 8, HS = 12)
    for (int i = threadIdx.x; i < plane_n; i += NT) plane[i] = 0ll;
    // scale of the fixed-point plane: sc = 2^(30 - ex) with max|grad_slots| < 2^ex over the chunk's units
    float gmax = 0.f;
    bool finite = true;
    for (int i = threadIdx.x; i < nq * Dh; i += NT) {
        const int qi = i / Dh, c = i - qi * Dh;
        const float v = fabsf(grad_slots[(((long long)b * Q + q0 + qi) * M + m) * Dh + c]);
        finite = finite && (v < __builtin_inff());             // false for inf and NaN
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
        int ex = (int)((gb >> 23) & 255u) - 126;               // gmax < 2^ex
        if (ex < -90) ex = -90;                                 // tiny gradients: keep both scales normal numbers
        const unsigned int sb = (unsigned int)(127 + 30 - ex) << 23, ib = (unsigned int)(127 - 30 + ex) << 23;
        __builtin_memcpy(&sc, &sb, 4);
        __builtin_memcpy(&inv_sc, &ib, 4);
    }
    for (int cam = 0; cam < Ncam; ++cam) {
        const long long bn = (long long)b * Ncam + cam;
        if (threadIdx.x == 0) *n_hits = 0;
        __syncthreads();
        for (int i0 = 0; i0 < nq; i0 += NT) {                 // ascending query order inside a wave's 64, waves in any order
            const int qi = i0 + threadIdx.x;
            bool hit = false;
            if (qi < nq) {
                const long long base = (((long long)cam * B + b) * Q + (q0 + qi)) * Za;
                for (int z = 0; z < Za; ++z) hit = hit || (mask[base + z] != 0);
            }
            const unsigned long long bal = __ballot(hit ? 1 : 0);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(n_hits, __popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (hit) hits[wbase + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = (unsigned short)qi;
        }
        __syncthreads();
        const int nh = *n_hits;
        for (int it = 0; it < nh; it += NT / 4) {
            const bool active = it + gidx < nh;
            if (__ballot(active ? 1 : 0) == 0ull) continue;
            const int q = q0 + (active ? (int)hits[it + gidx] : 0);
            const long long bq = (long long)b * Q + q;
            const long long u = bq * M + m;
            const long long base = (((long long)cam * B + b) * Q + q) * Za;
            // number of cameras that see the query: lane k tests cameras k, k+4, ...
            int count = 0;
            for (int c2 = k; c2 < Ncam; c2 += 4) {
                const long long b2 = (((long long)c2 * B + b) * Q + q) * Za;
                bool h2 = false;
                for (int z = 0; z < Za; ++z) h2 = h2 || (mask[b2 + z] != 0);
                count += h2 ? 1 : 0;
            }
            count += __shfl_xor(count, 1, 64);
            count += __shfl_xor(count, 2, 64);
            const float inv = (float)(count > 1 ? count : 1);
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (active && 4 * k + e < Dh) ? grad_slots[u * Dh + 4 * k + e] / inv : 0.f;
            // this lane's anchors: z = k and z = k + 4
            float rxo[2], ryo[2], dwo[2], ddwo[2];
            int bino[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int z = k + 4 * h;
                rxo[h] = ryo[h] = dwo[h] = ddwo[h] = 0.f;
                bino[h] = 0;
                if (active && z < Za) {
                    rxo[h] = ref_cam[(base + z) * 2];
                    ryo[h] = ref_cam[(base + z) * 2 + 1];
                    float fb = floorf(__fdiv_rn(__fsub_rn(qdepth[base + z], d0), dstep));
                    fb = fminf(fmaxf(fb, 0.f), (float)(DC - 1));
                    bino[h] = (int)fb;
                    dwo[h] = fbbev_plane_sample(pred_depth + (bn * DC + bino[h]) * (long long)(H0 * W0), H0, W0, rxo[h], ryo[h]);
                }
            }
            // sample lp of the unit: (B,Q,M,L,P[,2]) -> u*LP + lp, head-minor (B,Q,L,P,M[,2]) -> (bq*LP + lp)*M + m.  The next
            // sample's offsets / weight are requested before this sample's value loads (one exposed latency per sample).
            const int LP = L * P;
            const long long wo0 = (head_minor & 1) ? bq * LP * M + m : u * LP, wa0 = (head_minor & 2) ? bq * LP * M + m : u * LP;
            const int wo_step = (head_minor & 1) ? M : 1, wa_step = (head_minor & 2) ? M : 1;
            fbbev_v2f o_next = *reinterpret_cast<const fbbev_v2f*>(offsets + wo0 * 2);
            float a_next = attn[wa0];
            int lp = 0;
            for (int l = 0; l < L; ++l) {
                const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                const int ls = (int)level_start[l];
                const float* vp = value + (bn * S + ls) * row_stride + lane_off;
                long long* pl = plane + ls * HS + 4 * k;
                for (int p = 0; p < P; ++p, ++lp) {
                    const int nlp = lp + 1 < LP ? lp + 1 : lp;
                    const fbbev_v2f o = o_next;
                    const float a = active ? a_next : 0.f;
                    o_next = *reinterpret_cast<const fbbev_v2f*>(offsets + (wo0 + (long long)nlp * wo_step) * 2);
                    a_next = attn[wa0 + (long long)nlp * wa_step];
                    const int z = p % Za;
                    const int src = gbase | (z & 3);
                    const float rxz = __shfl(z < 4 ? rxo[0] : rxo[1], src, 64);
                    const float ryz = __shfl(z < 4 ? ryo[0] : ryo[1], src, 64);
                    const float dwz = __shfl(z < 4 ? dwo[0] : dwo[1], src, 64);
                    float h_im = -2.f, w_im = -2.f;
                    if (active) {
                        const float loc_w = rxz + __fdiv_rn(o[0], (float)sw);
                        const float loc_h = ryz + __fdiv_rn(o[1], (float)sh);
                        h_im = loc_h * sh - 0.5f;
                        w_im = loc_w * sw - 0.5f;
                    }
                    const bool inr = active && h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw;
                    const float weight = a * dwz;
                    float dot = 0.f, gx = 0.f, gy = 0.f;
                    if (inr && chunk_live) {
                        const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, 1);      // o1..o4 = token indices
                        const fbbev_v4f zero = {0.f, 0.f, 0.f, 0.f};
                        const bool k1 = s.o1 >= 0, k2 = s.o2 >= 0, k3 = s.o3 >= 0, k4 = s.o4 >= 0;
                        fbbev_v4f v1 = *reinterpret_cast<const fbbev_v4f*>(vp + (long long)(k1 ? s.o1 : 0) * row_stride);
                        fbbev_v4f v2 = *reinterpret_cast<const fbbev_v4f*>(vp + (long long)(k2 ? s.o2 : 0) * row_stride);
                        fbbev_v4f v3 = *reinterpret_cast<const fbbev_v4f*>(vp + (long long)(k3 ? s.o3 : 0) * row_stride);
                        fbbev_v4f v4 = *reinterpret_cast<const fbbev_v4f*>(vp + (long long)(k4 ? s.o4 : 0) * row_stride);
                        v1 = k1 ? v1 : zero; v2 = k2 ? v2 : zero; v3 = k3 ? v3 : zero; v4 = k4 ? v4 : zero;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float tgv = g[e] * weight * sc;                 // sc is a power of two: exact
                            const bool ce = 4 * k + e < Dh;                       // padding channels stay out of the atomics
                            if (k1 && ce) fbbev_lds_atomic_add_i64(pl + s.o1 * HS + e, (long long)__float2int_rn(s.w1 * tgv));
                            if (k2 && ce) fbbev_lds_atomic_add_i64(pl + s.o2 * HS + e, (long long)__float2int_rn(s.w2 * tgv));
                            if (k3 && ce) fbbev_lds_atomic_add_i64(pl + s.o3 * HS + e, (long long)__float2int_rn(s.w3 * tgv));
                            if (k4 && ce) fbbev_lds_atomic_add_i64(pl + s.o4 * HS + e, (long long)__float2int_rn(s.w4 * tgv));
                            dot += g[e] * (s.w1 * v1[e] + s.w2 * v2[e] + s.w3 * v3[e] + s.w4 * v4[e]);
                            gy += g[e] * (-s.hw * v1[e] - s.lw * v2[e] + s.hw * v3[e] + s.lw * v4[e]);
                            gx += g[e] * (-s.hh * v1[e] + s.hh * v2[e] - s.lh * v3[e] + s.lh * v4[e]);
                        }
                    }
                    dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64);
                    gx += __shfl_xor(gx, 1, 64);   gx += __shfl_xor(gx, 2, 64);
                    gy += __shfl_xor(gy, 1, 64);   gy += __shfl_xor(gy, 2, 64);
                    // the unit's weight / offset gradients of this camera: parked in LDS, added to global memory after the
                    // sample loop (a read-modify-write here would put a second global round trip into every sample)
                    if (k < 3) stage[(gidx * LP + lp) * 3 + k] = !inr ? 0.f : (k == 0 ? dwz * dot : (k == 1 ? weight * gx : weight * gy));
                    if (inr && k == (z & 3)) ddwo[z >> 2] += a * dot;
                }
            }
            // one add per camera that sees the query, in camera order: this workgroup is the unit's only writer and its
            // camera phases are separated by barriers.  The group's 4 lanes share the L*P samples; (a zero is not added:
            // the sample was outside the image)
            (void)__ballot(1);       // the wave's LDS writes above precede these reads (program order of one wave; this
                                     // makes the CPU emulator's lanes meet here as well)
            if (active) {
                for (int j = k; j < LP; j += 4) {
                    const float* st = stage + (gidx * LP + j) * 3;
                    const long long wo = wo0 + (long long)j * wo_step, wa = wa0 + (long long)j * wa_step;
                    const float sa = st[0], sx = st[1], sy = st[2];
                    if (sa != 0.f) grad_attn[wa] += sa;
                    if (sx != 0.f) grad_offsets[wo * 2] += sx;
                    if (sy != 0.f) grad_offsets[wo * 2 + 1] += sy;
                }
            }
            // dw[z] -> the four corners of the query's bin plane (fbbev_plane_sample), each anchor by its own lane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int z = k + 4 * h;
                if (!active || z >= Za || ddwo[h] == 0.f) continue;
                const float h_im = ryo[h] * H0 - 0.5f, w_im = rxo[h] * W0 - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H0 && w_im < (float)W0)) continue;
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, H0, W0, 1);
                float* gd = grad_pred_depth + (bn * DC + bino[h]) * (long long)(H0 * W0);
                if (s.o1 >= 0) fbbev_atomic_add_f32(gd + s.o1, s.w1 * ddwo[h]);
                if (s.o2 >= 0) fbbev_atomic_add_f32(gd + s.o2, s.w2 * ddwo[h]);
                if (s.o3 >= 0) fbbev_atomic_add_f32(gd + s.o3, s.w3 * ddwo[h]);
                if (s.o4 >= 0) fbbev_atomic_add_f32(gd + s.o4, s.w4 * ddwo[h]);
            }
        }
        __syncthreads();
        // the camera's plane -> this workgroup's slice of the partial buffer; cleared for the next camera on the way
        float* dst = part + ((((long long)b * M + m) * n_chunks + chunk) * Ncam + cam) * (long long)plane_n;
        for (int i = threadIdx.x * 4; i < plane_n; i += NT * 4) {
            fbbev_v4f t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[e] = poisoned ? __builtin_nanf("") : (float)plane[i + e] * inv_sc;      // one rounding (int64 -> fp32)
                plane[i + e] = 0ll;
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
