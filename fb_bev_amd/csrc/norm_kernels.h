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

// Backward of the above (training; torch's native_layer_norm_backward + its two gamma/beta kernels take 0.47 ms per
// LayerNorm on 160 000 x 80 -- 51 MB in, 51 MB out, i.e. ~12x the time the bytes need).  Half a wave64 owns a row, as in the
// forward; a workgroup walks rows with a grid stride.  Per row (g = dy * weight, xh = (x - mean) * inv):
//     dx = inv * (g - mean(g) - xh * mean(g * xh))            (the two means: half-wave shuffle reductions)
// and each lane keeps running sums of dy * xh and dy for its 4 channels; at the end the 8 half-waves of the workgroup meet
// in LDS (fixed order) and the workgroup writes ONE partial row pair: partial[wg][0][C] = sum dy * xh, partial[wg][1][C] =
// sum dy.  The caller sums the partial rows (deterministic: no atomics anywhere).  mean / inv are recomputed from x with the
// forward's own expression sequence.
__global__ void __launch_bounds__(256)
k_layernorm_rows_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ weight, float eps,
                     long long rows, int C, float* __restrict__ dx, float* __restrict__ partial) {
    __shared__ float red[8][2][128];
    const int lane = threadIdx.x & 63, l = lane & 31, hw = threadIdx.x >> 5;
    const bool chan = l * 4 < C;
    fbbev_v4f w = {0.f, 0.f, 0.f, 0.f};
    if (chan) w = *reinterpret_cast<const fbbev_v4f*>(weight + l * 4);
    fbbev_v4f sw = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * 8;
    const long long n_it = (rows + stride - 1) / stride;         // the same trip count for every lane (shuffles inside)
    for (long long it = 0; it < n_it; ++it) {
        const long long row = it * stride + (long long)blockIdx.x * 8 + hw;
        const bool act = chan && row < rows;
        fbbev_v4f v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            v = *reinterpret_cast<const fbbev_v4f*>(x + row * C + l * 4);
            g = *reinterpret_cast<const fbbev_v4f*>(dy + row * C + l * 4);
        }
        float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)C;
        fbbev_v4f d = {0.f, 0.f, 0.f, 0.f};
        if (act) { d[0] = v[0] - mean; d[1] = v[1] - mean; d[2] = v[2] - mean; d[3] = v[3] - mean; }
        float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float inv = 1.0f / sqrtf(q / (float)C + eps);
        fbbev_v4f xh, gw;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[e] = d[e] * inv;
            gw[e] = g[e] * w[e];
            a += gw[e];
            b += gw[e] * xh[e];
            sw[e] += g[e] * xh[e];
            sb[e] += g[e];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        a /= (float)C;
        b /= (float)C;
        if (act) {
            fbbev_v4f r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = inv * (gw[e] - a - xh[e] * b);
            *reinterpret_cast<fbbev_v4f*>(dx + row * C + l * 4) = r;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[hw][0][l * 4 + e] = sw[e]; red[hw][1][l * 4 + e] = sb[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) t += red[h][k][c];
        partial[((long long)blockIdx.x * 2 + k) * C + c] = t;
    }
}
