// x3_split.h -- the operand split of the "bf16 x 3" arithmetic (history_conv_x3_kernels.h, rows_linear_kernels.h, wgrad_kernels.h):
// v = hi + lo with hi = bf16(v), lo = bf16(v - hi); a product is three MFMAs a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, fp32 accumulation.
#pragma once
#include "rt.h"

// v (8 floats as two float4) -> hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void fbbev_split_bf16x8(const fbbev_v4f& lo4, const fbbev_v4f& hi4, fbbev_bf16x8& h, fbbev_bf16x8& l) {
    h = fbbev_cvt_bf16x8(lo4, hi4);
    fbbev_v4u u;
    __builtin_memcpy(&u, &h, 16);
    fbbev_v4f dl, dh;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned int a0 = u[e] << 16, a1 = u[e] & 0xffff0000u, b0 = u[2 + e] << 16, b1 = u[2 + e] & 0xffff0000u;
        float f0, f1, g0, g1;
        __builtin_memcpy(&f0, &a0, 4); __builtin_memcpy(&f1, &a1, 4); __builtin_memcpy(&g0, &b0, 4); __builtin_memcpy(&g1, &b1, 4);
        dl[2 * e] = lo4[2 * e] - f0; dl[2 * e + 1] = lo4[2 * e + 1] - f1;
        dh[2 * e] = hi4[2 * e] - g0; dh[2 * e + 1] = hi4[2 * e + 1] - g1;
    }
    l = fbbev_cvt_bf16x8(dl, dh);
}
