// norm_kernels.h -- LayerNorm over short channel rows (the three `norm` steps of BEVFormerEncoderLayer,
// bevformer_encoder.py:250-377 operation_order; mmcv build_norm_layer('LN') = torch.nn.LayerNorm).
// At FB-OCC sizes the rows are 80 floats and there are 10^4..10^5 of them per sample; torch's generic kernel spends
// ~150 us on 160k x 80.  Here half a wave64 owns a row (C/4 <= 32 lanes, one float4 each): mean and the biased
// variance are two shuffle reductions inside the half-wave (two-pass, as torch's CPU/GPU kernels: var = E[(x-mean)^2]),
// y = (x - mean) * rsqrt-free 1/sqrt(var + eps) * weight + bias.  Optional residual: y = LN(x + r).
// Bound: HBM (read + write once).
#pragma once
#include "rt.h"

__global__ void __launch_bounds__(256)
k_layernorm_rows(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ weight,
                 const float* __restrict__ bias, float eps, long long rows, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool live = row < rows;
    const bool act = live && l * 4 < C;
    fbbev_v4f v = {0.f, 0.f, 0.f, 0.f};
    if (act) {
        v = *reinterpret_cast<const fbbev_v4f*>(x + row * C + l * 4);
        if (r) {
            const fbbev_v4f t = *reinterpret_cast<const fbbev_v4f*>(r + row * C + l * 4);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
    }
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (act) { d0 = v[0] - mean; d1 = v[1] - mean; d2 = v[2] - mean; d3 = v[3] - mean; }
    float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float inv = 1.0f / sqrtf(q / (float)C + eps);
    if (act) {
        const fbbev_v4f w = *reinterpret_cast<const fbbev_v4f*>(weight + l * 4);
        const fbbev_v4f b = *reinterpret_cast<const fbbev_v4f*>(bias + l * 4);
        fbbev_v4f y;
        y[0] = d0 * inv * w[0] + b[0]; y[1] = d1 * inv * w[1] + b[1];
        y[2] = d2 * inv * w[2] + b[2]; y[3] = d3 * inv * w[3] + b[3];
        *reinterpret_cast<fbbev_v4f*>(out + row * C + l * 4) = y;
    }
}
