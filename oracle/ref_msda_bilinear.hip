// oracle/ref_msda_bilinear.hip -- TEST INFRASTRUCTURE ONLY (nothing under fb_bev_amd/ loads the result).
//
// mmcv's ms_deform_attn CUDA source is not in the FB-BEV tree, but the tree carries a twin of its two bilinear device
// functions: mmdet3d/ops/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh -- dcnv3_im2col_bilinear (:32-80) and
// dcnv3_col2im_bilinear (:82-147) are Deformable-DETR's ms_deform_attn_im2col_bilinear / _col2im_bilinear with `group`
// for `nheads` and one `offset_scale` factor where the original multiplies by width / height.  This file compiles that
// header FROM WHERE IT LIES (never copied) with the functions callable on the host, and exports them, so the MSDA part
// of the oracle is checked against reference-tree code for: corner selection, the per-corner zero padding, the bilinear
// weights, and the gradient expressions (value atomics, d/dw, d/dh, d/dweight).  The loops over levels / points /
// heads around them and the loc -> pixel mapping (x = loc*W - 0.5) remain a restatement of mmcv.
//   hipcc -x hip -include hip/hip_runtime.h -Ioracle/ref_shims -I<reference>/mmdet3d/ops/ops_dcnv3/src/cuda ...
#include <cmath>
#include <cstdio>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
// host-side overload of the one device-only call the two functions make
__attribute__((host)) inline float atomicAdd(float* p, float v) { const float o = *p; *p += v; return o; }
#define cudaStream_t hipStream_t
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
#include "dcnv3_im2col_cuda.cuh"

extern "C" float ref_im2col_bilinear(const float* data, int height, int width, int heads, int channels, float h, float w,
                                     int m, int c) {
    return dcnv3_im2col_bilinear<float>(data, height, width, heads, channels, h, w, m, c);
}

// grad_im (height*width*heads*channels) is accumulated into; grad_offset[2] = offset_scale * (d/dw, d/dh) * top_grad * mask;
// grad_mask[1] = top_grad * sample
extern "C" void ref_col2im_bilinear(const float* data, int height, int width, int heads, int channels, float h, float w,
                                    int m, int c, float offset_scale, float top_grad, float mask, float* grad_im,
                                    float* grad_offset, float* grad_mask) {
    dcnv3_col2im_bilinear<float>(data, height, width, heads, channels, h, w, m, c, offset_scale, top_grad, mask, grad_im,
                                 grad_offset, grad_mask);
}
