// Round-4 experiment, NOT part of libfbbev_hip.so: does a HEAD-PLANE layout of the camera tokens cut the vector-L1 accesses of the
// unit-per-lane bilinear sampler?  (the product sampler is bound by TCP accesses, ~43 lines per load instruction.)
//
//   token rows today   : [token][chunk k][head m][4 floats]  -- a (token, head) piece is 16 bytes inside a 384-byte row; two lanes
//                        share a line only when they sample the SAME token.
//   head planes (here) : [head m][token (y, x)][TS floats]   -- the tokens of one head are contiguous: x-neighbours are TS*4 bytes
//                        apart, so (a) the two x-corners of a sample are ONE contiguous run of 2*TS floats (5 or 6 sixteen-byte
//                        loads instead of 6 scattered ones), (b) lanes that sample neighbouring tokens of the same head share lines.
//   TS = 12 (16-byte aligned, 2 padding floats per token) or TS = 10 (dense, runs are 8-byte aligned).
// Mappings of lanes to (query, head) units: 4 heads x a 4x4 patch of BEV queries per wave (the product's), or ONE head x an 8x8 patch.
// All kernels run the product's sampler arithmetic; padded corners get weight 0 instead of reading zeros (same sums: w*0 == 0*v).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fb_bev_amd/csrc/hip_rt -I fb_bev_amd/csrc tools/micro/plane_sampler.hip \
//         -o tools/micro/plane_sampler && tools/micro/plane_sampler [Q] [LP] [coherent] [H] [W]
#include "rt.h"
#include "msda_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

constexpr int DH = 10, M = 8, HS = 12;

struct sample_in { float ox, oy, attn; };

// ---- reference: the product's layout and 4-heads x 4x4-patch mapping (one sample at a time)
__global__ void __launch_bounds__(256)
k_rows44(const float* __restrict__ value, const float* __restrict__ ref, const sample_in* __restrict__ smp, int units, int LP,
         int H, int W, int side, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int patches_x = side / 8;
    const int patch = blockIdx.x;
    const int m = 4 * (wave & 1) + (lane & 3), qi = lane >> 2;
    const int qx = (patch % patches_x) * 8 + (wave >> 1) * 4 + (qi & 3), qy = (patch / patches_x) * 4 + (qi >> 2);
    const int q = qy * side + qx;
    if (qy >= side || q * M + m >= units) return;
    const int unit = q * M + m;
    const int row_stride = M * HS, chunk_stride = M * 4;
    const unsigned lane_off = (unsigned)(m * 4 * 4);
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float col[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) col[c] = 0.f;
    const sample_in* sp = smp + unit;
    for (int i = 0; i < LP; ++i) {
        const sample_in s0 = sp[(long long)i * units];
        const float h_im = (ry + s0.oy / H) * H - 0.5f, w_im = (rx + s0.ox / W) * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const fbbev_bilinear s = fbbev_bilinear_setup(h_im, w_im, H, W, row_stride);
            fbbev_unit_sample<DH, 4>(value, lane_off, s, chunk_stride, s0.attn, col);
        }
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) out[(long long)unit * DH + c] = col[c];
}

// ---- head planes.  A sample = two row runs (y0, y0+1) of 2 tokens each; clamped so that every load stays inside the level:
//   x0 == -1     -> run starts at x = 0: the valid corner (x = 0) sits in SLOT 0 with the high-x weight, slot 1 gets weight 0
//   x0 == W - 1  -> run starts at x = W - 2: the valid corner sits in slot 1 with the low-x weight, slot 0 gets weight 0
//   y0 == -1 / y0 == H - 1 -> the invalid row reads the valid one again (same lines) with weight 0
struct plane_setup {
    unsigned r0, r1;                 // float offsets of the two runs inside the head plane
    float w00, w01, w10, w11;        // weights of (row 0 slot 0), (row 0 slot 1), (row 1 slot 0), (row 1 slot 1)
};
template <int TS>
__device__ __forceinline__ plane_setup plane_bilinear(float h, float w, int H, int W) {
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const float lh = h - (float)h_low, lw = w - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const bool left = w_low < 0, right = w_low >= W - 1;
    const int xb = left ? 0 : (right ? W - 2 : w_low);
    // slot weights along x: normally (hw, lw); at the left edge the x = 0 corner is the HIGH corner -> (lw, 0); at the right edge
    // the x = W-1 corner is the LOW corner -> (0, hw)
    const float sx0 = left ? lw : (right ? 0.f : hw), sx1 = left ? 0.f : (right ? hw : lw);
    const bool top = h_low < 0, bottom = h_low >= H - 1;
    const int y0 = top ? 0 : h_low, y1 = bottom ? h_low : h_low + 1;
    const float sy0 = top ? 0.f : hh, sy1 = bottom ? 0.f : lh;
    plane_setup s;
    s.r0 = (unsigned)((y0 * W + xb) * TS);
    s.r1 = (unsigned)((y1 * W + xb) * TS);
    s.w00 = sy0 * sx0; s.w01 = sy0 * sx1; s.w10 = sy1 * sx0; s.w11 = sy1 * sx1;
    return s;
}

// run of 2 tokens = 2*TS floats from a 4-byte aligned float pointer, as 16-byte loads (+ one 8-byte tail when 2*TS % 4 == 2 -- not here)
template <int TS>
struct run_regs { fbbev_v4f v[(2 * TS) / 4]; };

template <int TS>
__device__ __forceinline__ void load_run(const float* __restrict__ p, run_regs<TS>& r) {
    static_assert((2 * TS) % 4 == 0, "2 tokens = whole 16-byte pieces");
#pragma unroll
    for (int k = 0; k < (2 * TS) / 4; ++k) {
        // TS = 10: 8-byte aligned 16-byte loads (global memory takes dword-aligned multi-dword accesses)
        fbbev_v4f t;
        __builtin_memcpy(&t, p + 4 * k, 16);
        r.v[k] = t;
    }
}
template <int TS>
__device__ __forceinline__ float run_get(const run_regs<TS>& r, int slot, int c) {
    const int i = slot * TS + c;
    return r.v[i >> 2][i & 3];
}

template <int TS>
__device__ __forceinline__ void plane_blend(const run_regs<TS>& a, const run_regs<TS>& b, const plane_setup& s, float weight, float (&col)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; ++c)
        col[c] += (s.w00 * run_get<TS>(a, 0, c) + s.w01 * run_get<TS>(a, 1, c) + s.w10 * run_get<TS>(b, 0, c) + s.w11 * run_get<TS>(b, 1, c)) * weight;
}

// MAP 0: wave = 4 heads x 4x4 patch (workgroup = 8 heads of an 8x4 patch); MAP 1: wave = 1 head x 8x8 patch (workgroup = 4 heads; 2 per patch)
template <int TS, int MAP>
__device__ __forceinline__ bool lane_unit(int side, int units, int& q, int& m) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int patches_x = side / 8;
    int qx, qy;
    if (MAP == 0) {
        const int patch = blockIdx.x, qi = lane >> 2;
        m = 4 * (wave & 1) + (lane & 3);
        qx = (patch % patches_x) * 8 + (wave >> 1) * 4 + (qi & 3); qy = (patch / patches_x) * 4 + (qi >> 2);
    } else {
        const int patch = blockIdx.x >> 1;
        m = (blockIdx.x & 1) * 4 + wave;
        qx = (patch % patches_x) * 8 + (lane & 7); qy = (patch / patches_x) * 8 + (lane >> 3);
    }
    q = qy * side + qx;
    return qy < side && q * M + m < units;
}

template <int TS, int MAP>
__global__ void __launch_bounds__(256)
k_plane(const float* __restrict__ planes, const float* __restrict__ ref, const sample_in* __restrict__ smp, int units, int LP,
        int H, int W, int side, float* __restrict__ out) {
    int q, m;
    if (!lane_unit<TS, MAP>(side, units, q, m)) return;
    const int unit = q * M + m;
    const float* plane = planes + (size_t)m * H * W * TS;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float col[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) col[c] = 0.f;
    const sample_in* sp = smp + unit;
    for (int i = 0; i < LP; ++i) {
        const sample_in s0 = sp[(long long)i * units];
        const float h_im = (ry + s0.oy / H) * H - 0.5f, w_im = (rx + s0.ox / W) * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const plane_setup s = plane_bilinear<TS>(h_im, w_im, H, W);
            run_regs<TS> a, b;
            load_run<TS>(plane + s.r0, a);
            load_run<TS>(plane + s.r1, b);
            plane_blend<TS>(a, b, s, s0.attn, col);
        }
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) out[(long long)unit * DH + c] = col[c];
}

// two samples in flight per lane (the product's pipelining): an out-of-image sample loads the plane's first tokens with weight 0
template <int TS>
struct plane_pending { run_regs<TS> a, b; plane_setup s; float weight; };

template <int TS, int MAP>
__global__ void __launch_bounds__(256, 2)
k_plane_pipe(const float* __restrict__ planes, const float* __restrict__ ref, const sample_in* __restrict__ smp, int units, int LP,
             int H, int W, int side, float* __restrict__ out) {
    int q, m;
    if (!lane_unit<TS, MAP>(side, units, q, m)) return;
    const int unit = q * M + m;
    const float* plane = planes + (size_t)m * H * W * TS;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float col[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) col[c] = 0.f;
    const sample_in* sp = smp + unit;
    auto params = [&](int i) { return sp[(long long)(i < LP ? i : LP - 1) * units]; };
    auto start = [&](const sample_in& s0, plane_pending<TS>& p) {
        const float h_im = (ry + s0.oy / H) * H - 0.5f, w_im = (rx + s0.ox / W) * W - 0.5f;
        const bool live = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        p.s = plane_bilinear<TS>(live ? h_im : 0.f, live ? w_im : 0.f, H, W);
        p.weight = live ? s0.attn : 0.f;
        load_run<TS>(plane + p.s.r0, p.a);
        load_run<TS>(plane + p.s.r1, p.b);
    };
    plane_pending<TS> pa, pb;
    sample_in q0 = params(0), q1 = params(1);
    start(q0, pa);
    for (int i = 0; i < LP; i += 2) {
        q0 = params(i + 2);
        fbbev_sched_fence();
        start(q1, pb);
        fbbev_sched_fence();
        plane_blend<TS>(pa.a, pa.b, pa.s, pa.weight, col);
        q1 = params(i + 3);
        fbbev_sched_fence();
        start(q0, pa);
        fbbev_sched_fence();
        if (i + 1 < LP) plane_blend<TS>(pb.a, pb.b, pb.s, pb.weight, col);
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) out[(long long)unit * DH + c] = col[c];
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; }

int main(int argc, char** argv) {
    const int Q = argc > 1 ? atoi(argv[1]) : 160000, LP = argc > 2 ? atoi(argv[2]) : 32;
    const int coherent = argc > 3 ? atoi(argv[3]) : 1;
    const int H = argc > 4 ? atoi(argv[4]) : 32, W = argc > 5 ? atoi(argv[5]) : 88;
    const int S = H * W, units = Q * M;
    std::vector<float> value((size_t)S * M * HS), ref((size_t)Q * 2), p12((size_t)M * S * 12 + 16), p10((size_t)M * S * 10 + 16);
    std::vector<sample_in> smp((size_t)units * LP);
    unsigned seed = 12345u;
    for (auto& v : value) v = frand(seed) - 0.5f;
    // chunk-major rows [token][k][m][4] -> planes [m][token][TS]
    for (int t = 0; t < S; ++t)
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < 12; ++c) {
                const float v = c < DH ? value[(size_t)t * M * HS + (c >> 2) * M * 4 + m * 4 + (c & 3)] : 0.f;
                p12[((size_t)m * S + t) * 12 + c] = v;
                if (c < 10) p10[((size_t)m * S + t) * 10 + c] = v;
            }
    const int side = (int)ceil(sqrt((double)Q));
    for (int q = 0; q < Q; ++q) { ref[2 * q] = ((q % side) + 0.5f) / side; ref[2 * q + 1] = ((q / side) + 0.5f) / side; }
    if (!coherent) {
        for (auto& s : smp) { s.ox = (frand(seed) - 0.5f) * 12.f; s.oy = (frand(seed) - 0.5f) * 12.f; s.attn = frand(seed) / LP; }
    } else {
        for (int i = 0; i < LP; ++i)
            for (int u = 0; u < units; ++u) {
                const int m = u % M;
                const float th = 6.2831853f * m / M, r = 1.f + (i % 8);
                sample_in& s = smp[(size_t)i * units + u];
                s.ox = cosf(th) * r + (frand(seed) - 0.5f) * 0.3f; s.oy = sinf(th) * r + (frand(seed) - 0.5f) * 0.3f; s.attn = frand(seed) / LP;
            }
    }
    float *dv, *d12, *d10, *dr, *o0, *o1; sample_in* ds;
    hipMalloc(&dv, value.size() * 4); hipMalloc(&d12, p12.size() * 4); hipMalloc(&d10, p10.size() * 4);
    hipMalloc(&dr, ref.size() * 4); hipMalloc(&ds, smp.size() * sizeof(sample_in));
    hipMalloc(&o0, (size_t)units * DH * 4); hipMalloc(&o1, (size_t)units * DH * 4);
    hipMemcpy(dv, value.data(), value.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d12, p12.data(), p12.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d10, p10.data(), p10.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dr, ref.data(), ref.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, smp.data(), smp.size() * sizeof(sample_in), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int b44 = ((side + 7) / 8) * ((side + 3) / 4), b88 = ((side + 7) / 8) * ((side + 7) / 8) * 2;
    const char* names[] = {"rows_4x4x4heads", "plane12_4x4x4heads", "plane12_8x8x1head", "plane10_4x4x4heads", "plane10_8x8x1head",
                           "plane12_8x8x1head_pipelined", "plane10_8x8x1head_pipelined", "plane12_4x4x4heads_pipelined", "plane10_4x4x4heads_pipelined"};
    const int NV = 9;
    float ms[NV];
    std::vector<float> h0((size_t)units * DH), h1((size_t)units * DH);
    double worst = 0.0, maxv = 0.0;
    for (int which = 0; which < NV; ++which) {
        hipMemset(o1, 0, (size_t)units * DH * 4);
        for (int it = 0; it < 23; ++it) {
            if (it == 3) hipEventRecord(e0);
            switch (which) {
            case 0: hipLaunchKernelGGL(k_rows44, dim3(b44), dim3(256), 0, 0, dv, dr, ds, units, LP, H, W, side, o0); break;
            case 1: hipLaunchKernelGGL((k_plane<12, 0>), dim3(b44), dim3(256), 0, 0, d12, dr, ds, units, LP, H, W, side, o1); break;
            case 2: hipLaunchKernelGGL((k_plane<12, 1>), dim3(b88), dim3(256), 0, 0, d12, dr, ds, units, LP, H, W, side, o1); break;
            case 3: hipLaunchKernelGGL((k_plane<10, 0>), dim3(b44), dim3(256), 0, 0, d10, dr, ds, units, LP, H, W, side, o1); break;
            case 4: hipLaunchKernelGGL((k_plane<10, 1>), dim3(b88), dim3(256), 0, 0, d10, dr, ds, units, LP, H, W, side, o1); break;
            case 5: hipLaunchKernelGGL((k_plane_pipe<12, 1>), dim3(b88), dim3(256), 0, 0, d12, dr, ds, units, LP, H, W, side, o1); break;
            case 6: hipLaunchKernelGGL((k_plane_pipe<10, 1>), dim3(b88), dim3(256), 0, 0, d10, dr, ds, units, LP, H, W, side, o1); break;
            case 7: hipLaunchKernelGGL((k_plane_pipe<12, 0>), dim3(b44), dim3(256), 0, 0, d12, dr, ds, units, LP, H, W, side, o1); break;
            default: hipLaunchKernelGGL((k_plane_pipe<10, 0>), dim3(b44), dim3(256), 0, 0, d10, dr, ds, units, LP, H, W, side, o1); break;
            }
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[which], e0, e1);
        ms[which] /= 20.f;
        if (which == 0) hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost);
        else {
            hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < h0.size(); ++i) {
                const double d = fabs((double)h0[i] - h1[i]);
                if (d > worst) worst = d;
                if (fabs(h0[i]) > maxv) maxv = fabs(h0[i]);
            }
        }
    }
    printf("{\"experiment\": \"unit sampler: token rows vs head planes\", \"Q\": %d, \"LP\": %d, \"level\": [%d, %d], \"coherent_offsets\": %d", Q, LP, H, W, coherent);
    for (int i = 0; i < NV; ++i) printf(", \"%s_ms\": %.4f", names[i], ms[i]);
    printf(", \"max_abs_diff_vs_rows\": %.3g, \"max_abs\": %.3g, \"hip_error\": %d}\n", worst, maxv, (int)hipGetLastError());
    return worst <= 1e-5 * maxv ? 0 : 1;
}
