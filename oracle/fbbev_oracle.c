/*
 * fbbev_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the two native ops on the FB-OCC view-transformation
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the shipped path (fb_bev_amd/) never does.
 *
 *   bev_pool_v2 forward   follows  mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:18-45
 *   bev_pool_v2 backward  follows  mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:64-118
 *   ms_deform_attn fwd/bwd: the algorithm lives in mmcv-full 1.5.2
 *     (mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh), which is NOT
 *     under /root/reference.  Its published algorithm (Deformable-DETR
 *     im2col / col2im with bilinear sampling, zero padding, align_corners=False)
 *     is restated here and anchored on the reference call sites
 *     backward_projection/bevformer_utils/multi_scale_deformable_attn_function.py:127-133,159-169
 *     and spatial_cross_attention_depth.py:584-595.  mmcv itself ships no vector,
 *     but the reference tree carries a twin of its two bilinear device functions
 *     (mmdet3d/ops/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-147): msda_bilinear and
 *     the corner block of oracle_msda_bwd below reproduce them BIT FOR BIT on
 *     tests/golden/msda_bilinear_ref.npz (tests/test_oracle_msda_ref.py; generated
 *     from that header compiled where it lies, `make -C oracle ref_msda`).  What
 *     remains a restatement of mmcv: the loops over levels / points / heads and
 *     x = loc*W - 0.5; cross-checked against F.grid_sample in tests/test_oracle_msda.py.
 *
 * Pinned against: the 4-point known-answer fixture of
 *   mmdet3d/ops/bev_pool_v2/bev_pool.py:144-175 (tests/golden/bev_pool_v2_known_answer.json)
 * and, on the GPU box, against oracle/_ref (the reference's own .cu compiled for gfx950).
 *
 * use_fma: the reference kernel's `psum += *cur_feat * *cur_depth` is contracted to
 * an FMA by nvcc (-fmad=true default) and by hipcc (-ffp-contract=fast default);
 * use_fma=1 restates that, use_fma=0 is the un-contracted bracket.  This file must
 * be compiled with -ffp-contract=off so the choice is explicit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float mad(float a, float b, float c, int use_fma) {
    if (use_fma) return fmaf(a, b, c);
    volatile float p = a * b;  /* volatile: forbid re-fusion */
    return p + c;
}

/* bev_pool_cuda.cu:18-45 -- one (interval, channel) per CUDA thread, serial sum */
void oracle_bev_pool_v2_fwd(int c, int n_intervals, const float* depth, const float* feat,
                            const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
                            const int* interval_starts, const int* interval_lengths, float* out,
                            int use_fma) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int index = 0; index < n_intervals; ++index) {
        int interval_start = interval_starts[index];
        int interval_length = interval_lengths[index];
        for (int cur_c = 0; cur_c < c; ++cur_c) {
            float psum = 0.f;
            for (int i = 0; i < interval_length; ++i) {
                float d = depth[ranks_depth[interval_start + i]];
                float f = feat[(int64_t)ranks_feat[interval_start + i] * c + cur_c];
                psum = mad(f, d, psum, use_fma);
            }
            out[(int64_t)ranks_bev[interval_start] * c + cur_c] = psum;
        }
    }
}

/* bev_pool_cuda.cu:64-118 -- one interval (over ranks_feat) per CUDA thread */
void oracle_bev_pool_v2_bwd(int c, int n_intervals, const float* out_grad, const float* depth,
                            const float* feat, const int* ranks_depth, const int* ranks_feat,
                            const int* ranks_bev, const int* interval_starts,
                            const int* interval_lengths, float* depth_grad, float* feat_grad,
                            int use_fma) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int idx = 0; idx < n_intervals; ++idx) {
        int interval_start = interval_starts[idx];
        int interval_length = interval_lengths[idx];
        /* :88-102 depth_grad[ranks_depth[k]] = sum_c out_grad[ranks_bev[k]*c+cc]*feat[ranks_feat[k]*c+cc] */
        for (int i = 0; i < interval_length; ++i) {
            const float* og = out_grad + (int64_t)ranks_bev[interval_start + i] * c;
            const float* ft = feat + (int64_t)ranks_feat[interval_start + i] * c;
            float grad_sum = 0.f;
            for (int cur_c = 0; cur_c < c; ++cur_c) grad_sum = mad(og[cur_c], ft[cur_c], grad_sum, use_fma);
            depth_grad[ranks_depth[interval_start + i]] = grad_sum;
        }
        /* :104-117 feat_grad[ranks_feat[start]*c+cc] = sum_k out_grad[ranks_bev[k]*c+cc]*depth[ranks_depth[k]] */
        for (int cur_c = 0; cur_c < c; ++cur_c) {
            float grad_sum = 0.f;
            for (int i = 0; i < interval_length; ++i) {
                float og = out_grad[(int64_t)ranks_bev[interval_start + i] * c + cur_c];
                float d = depth[ranks_depth[interval_start + i]];
                grad_sum = mad(og, d, grad_sum, use_fma);
            }
            feat_grad[(int64_t)ranks_feat[interval_start] * c + cur_c] = grad_sum;
        }
    }
}

/* ---- multi-scale deformable attention (mmcv-full 1.5.2 semantics, restated) ---- */

/* bilinear sample of value[(h*W + w), m, c] with per-corner zero padding */
static float msda_bilinear(const float* bottom, int height, int width, int nheads, int channels,
                           float h, float w, int m, int c) {
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
    int w_stride = nheads * channels, h_stride = width * w_stride;
    int base = m * channels + c;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = bottom[h_low * h_stride + w_low * w_stride + base];
    if (h_low >= 0 && w_high <= width - 1) v2 = bottom[h_low * h_stride + w_high * w_stride + base];
    if (h_high <= height - 1 && w_low >= 0) v3 = bottom[h_high * h_stride + w_low * w_stride + base];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = bottom[h_high * h_stride + w_high * w_stride + base];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* value (B,S,M,Dh) ; spatial_shapes (L,2) int64 (h,w) ; level_start (L) int64 ;
 * loc (B,Q,M,L,P,2) (x,y) in [0,1] ; w (B,Q,M,L,P) ; out (B,Q,M*Dh) */
void oracle_msda_fwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                     const float* loc, const float* attn, int batch, int spatial_size, int num_heads,
                     int channels, int num_levels, int num_query, int num_point, float* out) {
    int64_t qid_stride = (int64_t)num_heads * channels;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b)
        for (int q = 0; q < num_query; ++q)
            for (int m = 0; m < num_heads; ++m) {
                int64_t sidx = ((int64_t)b * num_query + q) * num_heads + m;
                for (int c = 0; c < channels; ++c) {
                    float col = 0.f;
                    int64_t wp = sidx * num_levels * num_point, lp = wp * 2;
                    for (int l = 0; l < num_levels; ++l) {
                        int sh = (int)spatial_shapes[2 * l], sw = (int)spatial_shapes[2 * l + 1];
                        const float* vptr = value + ((int64_t)b * spatial_size + level_start[l]) * qid_stride;
                        for (int p = 0; p < num_point; ++p) {
                            float loc_w = loc[lp], loc_h = loc[lp + 1], weight = attn[wp];
                            float h_im = loc_h * sh - 0.5f, w_im = loc_w * sw - 0.5f;
                            if (h_im > -1 && w_im > -1 && h_im < sh && w_im < sw)
                                col += msda_bilinear(vptr, sh, sw, num_heads, channels, h_im, w_im, m, c) * weight;
                            wp += 1; lp += 2;
                        }
                    }
                    out[sidx * channels + c] = col;
                }
            }
}

/* grads accumulate into pre-zeroed grad_value / grad_loc / grad_attn (as the reference wrapper
 * provides, multi_scale_deformable_attn_function.py:155-157) */
void oracle_msda_bwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                     const float* loc, const float* attn, const float* grad_out, int batch,
                     int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                     int num_point, float* grad_value, float* grad_loc, float* grad_attn) {
    int64_t qid_stride = (int64_t)num_heads * channels;
    for (int b = 0; b < batch; ++b)
        for (int q = 0; q < num_query; ++q)
            for (int m = 0; m < num_heads; ++m) {
                int64_t sidx = ((int64_t)b * num_query + q) * num_heads + m;
                for (int c = 0; c < channels; ++c) {
                    float top_grad = grad_out[sidx * channels + c];
                    int64_t wp = sidx * num_levels * num_point, lp = wp * 2;
                    for (int l = 0; l < num_levels; ++l) {
                        int height = (int)spatial_shapes[2 * l], width = (int)spatial_shapes[2 * l + 1];
                        int64_t voff = ((int64_t)b * spatial_size + level_start[l]) * qid_stride;
                        const float* bottom = value + voff;
                        float* gv = grad_value + voff;
                        for (int p = 0; p < num_point; ++p, wp += 1, lp += 2) {
                            float loc_w = loc[lp], loc_h = loc[lp + 1], weight = attn[wp];
                            float h = loc_h * height - 0.5f, w = loc_w * width - 0.5f;
                            if (!(h > -1 && w > -1 && h < height && w < width)) continue;
                            int h_low = (int)floorf(h), w_low = (int)floorf(w);
                            int h_high = h_low + 1, w_high = w_low + 1;
                            float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
                            int w_stride = num_heads * channels, h_stride = width * w_stride;
                            int base = m * channels + c;
                            float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                            float tgv = top_grad * weight;
                            float ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                            if (h_low >= 0 && w_low >= 0) {
                                int o = h_low * h_stride + w_low * w_stride + base;
                                v1 = bottom[o]; ghw -= hw * v1; gww -= hh * v1; gv[o] += w1 * tgv;
                            }
                            if (h_low >= 0 && w_high <= width - 1) {
                                int o = h_low * h_stride + w_high * w_stride + base;
                                v2 = bottom[o]; ghw -= lw * v2; gww += hh * v2; gv[o] += w2 * tgv;
                            }
                            if (h_high <= height - 1 && w_low >= 0) {
                                int o = h_high * h_stride + w_low * w_stride + base;
                                v3 = bottom[o]; ghw += hw * v3; gww -= lh * v3; gv[o] += w3 * tgv;
                            }
                            if (h_high <= height - 1 && w_high <= width - 1) {
                                int o = h_high * h_stride + w_high * w_stride + base;
                                v4 = bottom[o]; ghw += lw * v4; gww += lh * v4; gv[o] += w4 * tgv;
                            }
                            float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                            grad_attn[wp] += top_grad * val;
                            grad_loc[lp] += width * gww * tgv;
                            grad_loc[lp + 1] += height * ghw * tgv;
                        }
                    }
                }
            }
}
