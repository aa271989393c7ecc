// msda_kernels.h -- multi-scale deformable attention sampling (forward / backward) for gfx950.
//
// Replaces mmcv._ext.ms_deform_attn_forward / ms_deform_attn_backward (mmcv-full 1.5.2; the
// source is NOT in the FB-BEV tree).  Call sites in the reference:
//   backward_projection/bevformer_utils/multi_scale_deformable_attn_function.py:127-133,159-169
//   backward_projection/bevformer_utils/spatial_cross_attention_depth.py:586-588 (depth map sampled
//     as a 1-head x D-channel value), :593-595 (value sampling, 8 heads x 10 channels).
// Semantics: out[b,q,m,c] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,c],
//   x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5), sample counted only if -1 < x < W_l and -1 < y < H_l,
//   the 4 corners individually zero-padded  (== F.grid_sample bilinear / zeros / align_corners=False).
//
// Mapping: forward = one lane per output scalar, channel fastest, so the Dh lanes of one
// (b,q,m) read Dh contiguous floats per corner and share (broadcast) the loc/weight loads.
// Backward = a group of GW lanes (16/32/64) per (b,q,m); channels strided over the group;
// d/dloc and d/dweight are reduced over channels with wave shuffles (no LDS), d/dvalue goes out
// through hardware fp32 atomics.
#pragma once
#include "rt.h"

struct fbbev_bilinear {
    int o1, o2, o3, o4;        // corner offsets (floats) relative to value_l + m*Dh + c; -1 = padded
    float w1, w2, w3, w4;      // corner weights
    float lh, lw, hh, hw;
};

__device__ __forceinline__ fbbev_bilinear fbbev_bilinear_setup(float h, float w, int height, int width,
                                                              int row_stride /* M*Dh */) {
    fbbev_bilinear s;
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    s.lh = h - (float)h_low; s.lw = w - (float)w_low; s.hh = 1.f - s.lh; s.hw = 1.f - s.lw;
    const int hs = width * row_stride;
    s.o1 = (h_low >= 0 && w_low >= 0) ? h_low * hs + w_low * row_stride : -1;
    s.o2 = (h_low >= 0 && w_high <= width - 1) ? h_low * hs + w_high * row_stride : -1;
    s.o3 = (h_high <= height - 1 && w_low >= 0) ? h_high * hs + w_low * row_stride : -1;
    s.o4 = (h_high <= height - 1 && w_high <= width - 1) ? h_high * hs + w_high * row_stride : -1;
    s.w1 = s.hh * s.hw; s.w2 = s.hh * s.lw; s.w3 = s.lh * s.hw; s.w4 = s.lh * s.lw;
    return s;
}

// One bilinear sample of a (b,q,head) unit's DH channels, accumulated into col[]: the body shared by the unit-per-lane
// kernels.  WIDE: corners are read as DHP/4 16-byte loads, `chunk_stride` floats apart (4 for head-major rows, M*4 for
// quad-interleaved ones).  The loads are unconditional -- a padded corner reads the level's first token instead and its
// floats are replaced by zeros afterwards -- so the 4*DHP/4 loads of a sample issue back to back instead of each
// sitting in its own exec-masked branch (what `valid ? *p : 0` compiles to); the arithmetic is unchanged.
template <int DH, int VEC>        // VEC = floats per load: 4 (WIDE), 2 (rows 8-byte aligned) or 1
__device__ __forceinline__ void fbbev_unit_sample(const float* __restrict__ value, unsigned lane_off,
                                                  const fbbev_bilinear& s, int chunk_stride, float weight,
                                                  float (&col)[DH]) {
    // value: the tensor's base (wave-uniform); lane_off: BYTE offset of this lane's head chunk in the level's first token
    // (the host guarantees the tensor is < 4 GiB: a uniform base + 32-bit lane offset is one address VGPR per load)
    constexpr int DHP = (DH + 3) / 4 * 4;
    const char* vb = reinterpret_cast<const char*>(value);
    const float* vp = reinterpret_cast<const float*>(vb + lane_off);
    float v1[DHP], v2[DHP], v3[DHP], v4[DHP];
    if constexpr (VEC == 4) {
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
#pragma unroll
        for (int k = 0; k < DHP / 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v1[4 * k + e] = k1 ? a1[k][e] : 0.f; v2[4 * k + e] = k2 ? a2[k][e] : 0.f;
                v3[4 * k + e] = k3 ? a3[k][e] : 0.f; v4[4 * k + e] = k4 ? a4[k][e] : 0.f;
            }
    } else if constexpr (VEC == 2) {
#pragma unroll
        for (int c = 0; c < DH; c += 2) {
            const fbbev_v2f zero = {0.f, 0.f};
            const fbbev_v2f a1 = s.o1 >= 0 ? *reinterpret_cast<const fbbev_v2f*>(vp + s.o1 + c) : zero;
            const fbbev_v2f a2 = s.o2 >= 0 ? *reinterpret_cast<const fbbev_v2f*>(vp + s.o2 + c) : zero;
            const fbbev_v2f a3 = s.o3 >= 0 ? *reinterpret_cast<const fbbev_v2f*>(vp + s.o3 + c) : zero;
            const fbbev_v2f a4 = s.o4 >= 0 ? *reinterpret_cast<const fbbev_v2f*>(vp + s.o4 + c) : zero;
            v1[c] = a1[0]; v1[c + 1] = a1[1]; v2[c] = a2[0]; v2[c + 1] = a2[1];
            v3[c] = a3[0]; v3[c + 1] = a3[1]; v4[c] = a4[0]; v4[c + 1] = a4[1];
        }
    } else {
#pragma unroll
        for (int c = 0; c < DH; ++c) {
            v1[c] = s.o1 >= 0 ? vp[s.o1 + c] : 0.f;
            v2[c] = s.o2 >= 0 ? vp[s.o2 + c] : 0.f;
            v3[c] = s.o3 >= 0 ? vp[s.o3 + c] : 0.f;
            v4[c] = s.o4 >= 0 ? vp[s.o4 + c] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) col[c] += (s.w1 * v1[c] + s.w2 * v2[c] + s.w3 * v3[c] + s.w4 * v4[c]) * weight;
}

// The same sample on 16-bit token rows (ET 1 = bf16, 2 = f16; fp32 accumulate): rows are chunk-major with EIGHT elements
// per (chunk, head) -- [chunk k][head m][8 x 16 bit] -- so a head's chunk is again one aligned 16-byte load and the 8 head
// lanes read one 128-byte line per instruction, for half the bytes per corner (Dh = 10: 16 + 4 bytes instead of 40).
// Elements are widened exactly; the arithmetic after the load is the fp32 kernel's.
template <int DH, int ET>
__device__ __forceinline__ void fbbev_unit_sample16(const void* __restrict__ value, unsigned lane_off /* bytes */,
                                                    const fbbev_bilinear& s, int chunk_stride /* elements */, float weight,
                                                    float (&col)[DH]) {
    constexpr int NCH = (DH + 7) / 8;
    const char* vb = reinterpret_cast<const char*>(value);
    const bool k1 = s.o1 >= 0, k2 = s.o2 >= 0, k3 = s.o3 >= 0, k4 = s.o4 >= 0;
    const unsigned b1 = lane_off + (k1 ? (unsigned)s.o1 * 2u : 0u), b2 = lane_off + (k2 ? (unsigned)s.o2 * 2u : 0u);
    const unsigned b3 = lane_off + (k3 ? (unsigned)s.o3 * 2u : 0u), b4 = lane_off + (k4 ? (unsigned)s.o4 * 2u : 0u);
    const unsigned cs = (unsigned)chunk_stride * 2u;
    fbbev_v4u a1[NCH], a2[NCH], a3[NCH], a4[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        a1[k] = *reinterpret_cast<const fbbev_v4u*>(vb + (b1 + k * cs));
        a2[k] = *reinterpret_cast<const fbbev_v4u*>(vb + (b2 + k * cs));
        a3[k] = *reinterpret_cast<const fbbev_v4u*>(vb + (b3 + k * cs));
        a4[k] = *reinterpret_cast<const fbbev_v4u*>(vb + (b4 + k * cs));
    }
    auto widen = [](unsigned int word, int half) -> float {
        if constexpr (ET == 2) return fbbev_f16_bits_to_f32(half ? (word >> 16) : (word & 0xffffu));
        else {
            const unsigned int u = half ? (word & 0xffff0000u) : (word << 16);
            float f;
            __builtin_memcpy(&f, &u, 4);
            return f;
        }
    };
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        const int k = c >> 3, wd = (c & 7) >> 1, hf = c & 1;
        const float v1 = k1 ? widen(a1[k][wd], hf) : 0.f, v2 = k2 ? widen(a2[k][wd], hf) : 0.f;
        const float v3 = k3 ? widen(a3[k][wd], hf) : 0.f, v4 = k4 ? widen(a4[k][wd], hf) : 0.f;
        col[c] += (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * weight;
    }
}

__global__ void __launch_bounds__(256)
k_msda_fwd(long long n, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
           const int64_t* __restrict__ level_start, const float* __restrict__ loc,
           const float* __restrict__ attn, int spatial_size, int M, int Dh, int L, int Q, int P,
           float* __restrict__ out) {
    const int row_stride = M * Dh;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Dh);
        const long long unit = idx / Dh;          // (b*Q + q)*M + m
        const int m = (int)(unit % M);
        const long long b = unit / M / Q;
        long long wp = unit * L * P, lp = wp * 2;
        float col = 0.f;
        for (int l = 0; l < L; ++l) {
            const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
            const float* vp = value + (b * spatial_size + level_start[l]) * row_stride + m * Dh + c;
            for (int p = 0; p < P; ++p, wp += 1, lp += 2) {
                const float loc_w = loc[lp], loc_h = loc[lp + 1], weight = attn[wp];
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
        out[idx] = col;
    }
}

// ---------------------------------------------------------------- BEV self-attention sampling, fused (inference)
// mmcv MultiScaleDeformableAttention.forward (external, mmcv/ops/multi_scale_deform_attn.py) builds
//   sampling_locations = reference_points[:, :, None, :, None, :] + sampling_offsets / offset_normalizer
// with two elementwise passes over the (B,Q,M,L,P,2) tensor and then calls ms_deform_attn_forward.  Here a lane owns a
// (b,q,head) unit (all DH channels, as k_da_cross_attn_fwd_unit): loc = ref + __fdiv_rn(offset, size) is evaluated in
// the kernel -- the same two correctly rounded fp32 operations -- the bilinear setup happens once per sample instead of
// once per channel lane, and with head-padded value rows (WIDE) a corner is DHP/4 aligned dwordx4 loads.
// ref (B,Q,L,2); offsets (B,Q,M,L,P,2) raw, or head-minor (B,Q,L,P,M,2) when off_head_minor; attn (B,Q,M,L,P) softmaxed.
template <int DH, bool WIDE, bool QI>
__global__ void __launch_bounds__(256)
k_msda_fwd_unit(long long n_units, const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
                const int64_t* __restrict__ level_start, const float* __restrict__ ref,
                const float* __restrict__ offsets, const float* __restrict__ attn, int spatial_size, int M, int L,
                int Q, int P, int HS, int off_head_minor, int stage_attn, float* __restrict__ out) {
    static_assert(!QI || WIDE, "quad-interleaved rows are read with 16-byte loads");
    // QI (value rows stored [chunk][head][4 floats]) and the LDS-staged attention weights: see k_da_cross_attn_fwd_unit
    const int head_off_m = QI ? 4 : HS, chunk_stride = QI ? M * 4 : 4;
    const int row_stride = M * HS;
    const int LP = L * P, LDW = LP + 1;
    float* staged = fbbev_dyn_lds_f32();          // [256][LP+1] when stage_attn
    // XCD-contiguous order: workgroup w runs on XCD w % 8 (each with its own L2); XCD x takes the x-th eighth of the
    // unit range -- a contiguous piece of the BEV plane, whose queries project into the same few camera regions -- instead
    // of every eighth workgroup of the whole plane (gridDim.x is a multiple of 8)
    const long long n_wg = (n_units + blockDim.x - 1) / blockDim.x, per_xcd = (n_wg + 7) / 8;
    for (long long w = blockIdx.x; (w >> 3) < per_xcd; w += gridDim.x) {
        const long long ubase = ((w & 7) * per_xcd + (w >> 3)) * blockDim.x;
        const long long unit = ubase + threadIdx.x;
        if (stage_attn) {
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
        const long long b = bq / Q;
        float col[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) col[c] = 0.f;
        // offsets of sample lp: (B,Q,M,L,P,2) -> unit*LP + lp, head-minor (B,Q,L,P,M,2) -> (bq*LP + lp)*M + m
        const fbbev_v2f* op = reinterpret_cast<const fbbev_v2f*>(offsets) + (off_head_minor ? bq * LP * M + m : unit * LP);
        const int wo_step = off_head_minor ? M : 1;
        fbbev_v2f o_next = op[0];
        int lp = 0;
        for (int l = 0; l < L; ++l) {
            const int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
            const float rx = ref[(bq * L + l) * 2], ry = ref[(bq * L + l) * 2 + 1];
            const unsigned lane_off = (unsigned)(((b * spatial_size + level_start[l]) * row_stride + m * head_off_m) * 4);
            for (int p = 0; p < P; ++p, ++lp) {
                // the next sample's offsets are requested before this sample's value loads: one exposed memory latency
                // per sample instead of two dependent ones
                const fbbev_v2f o = o_next;
                o_next = op[(long long)(lp + 1 < LP ? lp + 1 : lp) * wo_step];
                const float loc_w = rx + __fdiv_rn(o[0], (float)sw), loc_h = ry + __fdiv_rn(o[1], (float)sh);
                float weight;                                   // (uniform branch, LDS read kept an LDS read: see k_da_cross_attn_fwd_unit)
                if (stage_attn) weight = fbbev_lds_ld_f32(my_attn + lp);
                else weight = attn[unit * LP + lp];
                const float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                if (!(h_im > -1.f && w_im > -1.f && h_im < (float)sh && w_im < (float)sw)) continue;
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, sh, sw, row_stride);
                fbbev_unit_sample<DH, WIDE ? 4 : 1>(value, lane_off, s, chunk_stride, weight, col);
            }
        }
        float* dst = out + unit * DH;
#pragma unroll
        for (int c = 0; c < DH; ++c) dst[c] = col[c];
    }
}

template <int GW>
__device__ __forceinline__ float fbbev_group_sum(float v) {
#pragma unroll
    for (int o = GW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SCATTER = false: the unit-owned gradients only (grad_value comes from k_msda_bwd_scatter, msda_bwd_kernels.h)
template <int GW, bool SCATTER = true>
__global__ void __launch_bounds__(256)
k_msda_bwd(long long n_units, const float* __restrict__ value,
           const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
           const float* __restrict__ loc, const float* __restrict__ attn,
           const float* __restrict__ grad_out, int spatial_size, int M, int Dh, int L, int Q, int P,
           float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn) {
    const int slot = threadIdx.x % GW;
    const long long unit = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / GW;
    const bool active = unit < n_units;
    const long long u = active ? unit : 0;
    const int m = (int)(u % M);
    const long long b = u / M / Q;
    const int row_stride = M * Dh;
    long long wp = u * L * P, lp = wp * 2;
    for (int l = 0; l < L; ++l) {
        const int height = (int)spatial_shapes[2 * l], width = (int)spatial_shapes[2 * l + 1];
        const long long voff = (b * spatial_size + level_start[l]) * row_stride + m * Dh;
        for (int p = 0; p < P; ++p, wp += 1, lp += 2) {
            const float loc_w = loc[lp], loc_h = loc[lp + 1], weight = attn[wp];
            const float h_im = loc_h * height - 0.5f, w_im = loc_w * width - 0.5f;
            const bool inr = active && h_im > -1.f && w_im > -1.f && h_im < (float)height &&
                             w_im < (float)width;
            float g_w = 0.f, g_x = 0.f, g_y = 0.f;
            if (inr) {
                const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, height, width, row_stride);
                for (int c = slot; c < Dh; c += GW) {
                    const float top = grad_out[u * Dh + c];
                    const float tgv = top * weight;
                    const float* vp = value + voff + c;
                    float* gp = grad_value + voff + c;
                    (void)gp;
                    float ghw = 0.f, gww = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                    if (s.o1 >= 0) { v1 = vp[s.o1]; ghw -= s.hw * v1; gww -= s.hh * v1; if constexpr (SCATTER) fbbev_atomic_add_f32(gp + s.o1, s.w1 * tgv); }
                    if (s.o2 >= 0) { v2 = vp[s.o2]; ghw -= s.lw * v2; gww += s.hh * v2; if constexpr (SCATTER) fbbev_atomic_add_f32(gp + s.o2, s.w2 * tgv); }
                    if (s.o3 >= 0) { v3 = vp[s.o3]; ghw += s.hw * v3; gww -= s.lh * v3; if constexpr (SCATTER) fbbev_atomic_add_f32(gp + s.o3, s.w3 * tgv); }
                    if (s.o4 >= 0) { v4 = vp[s.o4]; ghw += s.lw * v4; gww += s.lh * v4; if constexpr (SCATTER) fbbev_atomic_add_f32(gp + s.o4, s.w4 * tgv); }
                    const float val = s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
                    g_w += top * val;
                    g_x += (float)width * gww * tgv;
                    g_y += (float)height * ghw * tgv;
                }
            }
            g_w = fbbev_group_sum<GW>(g_w);
            g_x = fbbev_group_sum<GW>(g_x);
            g_y = fbbev_group_sum<GW>(g_y);
            if (active && slot == 0) {
                grad_attn[wp] += g_w;
                grad_loc[lp] += g_x;
                grad_loc[lp + 1] += g_y;
            }
        }
    }
}
