// conv3d_kernels.h -- dense 3-D convolutions of FB-OCC's voxel encoder / occupancy head as fp32-MFMA implicit GEMMs
// (inference).
//
// Replaces, for eval-mode modules, the Conv3d(+BatchNorm)(+residual)(+ReLU) groups of
//   CustomResNet3D   mmdet3d/models/fbbev/modules/resnet3d.py:19-43 (conv3x3x3 / conv1x1x1), :78-102 (BasicBlock)
//   FPN3D            mmdet3d/models/fbbev/modules/fpn3d.py:50-70 (ConvModule 1x1x1 / 3x3x3)
//   OccHead          mmdet3d/models/fbbev/heads/occupancy_head.py:82-141 (3x3x3, 1x1x1, deconv3d k=2 s=2)
// which the reference runs in fp32 (@force_fp32) through the vendor library.  Measured on MI355X
// (profiles/r01_time_full.jsonl): those stacks are 33 of the 41.7 ms of a frame -- 670 GFLOP at ~20 TFLOP/s.
//
// Layout: activations NDHWC (torch channels_last_3d), so the Cin floats of a voxel are contiguous.  Batch norm is folded
// into the weights / bias on the host; bias, residual add and ReLU run in the store epilogue.
//
// GEMM view: D[cout][voxel] = sum over (tap, cin) of W[cout][tap][cin] * X[voxel + tap][cin], on
// v_mfma_f32_16x16x4_f32 (exact fp32; fragment layouts as in history_conv_kernels.h, validated on the hardware there).
// A wave owns NT=4 voxel tiles (64 consecutive output voxels, flat (b,d,h,w) order) x MT cout tiles; a workgroup is 4
// waves with the same cout tiles (their weight fragments hit the same L1 lines) and 256 consecutive voxels.
// K is walked tap by tap, 16 input channels at a time: lane (voxel i = lane%16, kk = lane/16) loads ONE float4
//   X[voxel_i + tap][16j + 4kk .. 4kk+3]        (a voxel's 4 lanes read 64 contiguous bytes)
// and uses element e of it as the B operand of k-step (j, e): the K order inside a 16-channel group is permuted
// (k-step (j,e) covers channels 16j + {e, 4+e, 8+e, 12+e}), which a dot product does not care about as long as the
// weights follow: the host stores them in fragment order
//   wf[tap][j][mt][lane][e] = W[cout = 16mt + lane%16][cin = 16j + 4(lane/16) + e][tap]
// so a lane's four A operands of a (tap, j, mt) are one aligned float4 and a wave's load is 1 KB contiguous.
// Per (tap, j): NT + MT float4 loads feed 4 * MT * NT MFMAs (64 at MT=4).  Zero padding = B operand 0.
// Transposed convolution k=2 s=2 (the head's `deblock`): every output voxel (2d+a, 2h+b, 2w+c) sees exactly one tap,
// i.e. 8 independent 1x1x1 convolutions with a strided store; the slowest grid index is the parity (a,b,c).
// Bound: fp32 MFMA (157 TFLOP/s peak).
#pragma once
#include "rt.h"

// KD = taps along the slowest spatial axis: KS for a 3-D convolution, 1 for a 2-D convolution on an NHWC image (= NDHWC
// with D = 1; no padding along that axis)
template <int KD, int KS, int MT>
__global__ void __launch_bounds__(256)
k_conv3d_ndhwc(const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bias,
               const float* __restrict__ residual, float* __restrict__ out, int B, int Di, int Hi, int Wi, int Cin,
               int Do, int Ho, int Wo, int Cout, int mt_total, int stride, int pad, int relu, int mode,
               long long wf_parity_stride, int gx, int gy) {
    constexpr int NT = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, i = lane & 15;
    const long long nvox = (long long)B * Do * Ho * Wo;
    // 1-D grid: voxel block fastest, then cout block, then (transposed only) output parity
    const int bx = blockIdx.x % gx, by = (blockIdx.x / gx) % gy, parity = blockIdx.x / (gx * gy);
    const long long base = ((long long)bx * 4 + wave) * (16 * NT);
    if (base >= nvox) return;                                // whole wave: there is no workgroup barrier below
    const int mt0 = by * MT;
    int bq[NT], dq[NT], hq[NT], wq[NT];
    bool vq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long n = base + 16 * t + i;
        vq[t] = n < nvox;
        long long r = vq[t] ? n : 0;
        wq[t] = (int)(r % Wo); r /= Wo;
        hq[t] = (int)(r % Ho); r /= Ho;
        dq[t] = (int)(r % Do);
        bq[t] = (int)(r / Do);
    }
    fbbev_v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    const int J = Cin >> 4;
    const float* __restrict__ wfp = wf + (long long)parity * wf_parity_stride + (long long)mt0 * 256 + lane * 4;
    const int pad_d = KD == 1 ? 0 : pad;
    for (int tap = 0; tap < KD * KS * KS; ++tap) {
        const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
        long long off[NT];
        bool ok[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int di, hi, wi;
            if (mode == 2) {
                // data gradient of a stride-s convolution: dx[i] = sum_k W_k^T dy[(i + pad - k) / s] over the taps for which
                // the division is exact -- `x` is dy here, (Di,Hi,Wi) its extent and the "output" voxel is the dx voxel
                const int nd = dq[t] + pad_d - kd, nh = hq[t] + pad - kh, nw = wq[t] + pad - kw;
                di = nd / stride; hi = nh / stride; wi = nw / stride;
                ok[t] = vq[t] && nd >= 0 && nh >= 0 && nw >= 0 && di * stride == nd && hi * stride == nh && wi * stride == nw &&
                        di < Di && hi < Hi && wi < Wi;
            } else {
                di = dq[t] * stride + kd - pad_d; hi = hq[t] * stride + kh - pad; wi = wq[t] * stride + kw - pad;
                ok[t] = vq[t] && di >= 0 && di < Di && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            }
            off[t] = (ok[t] ? ((((long long)bq[t] * Di + di) * Hi + hi) * Wi + wi) * Cin : 0) + 4 * g;
        }
        const float* __restrict__ wt = wfp + (long long)tap * J * mt_total * 256;
        // ping-pong register buffers over the 16-channel groups of this tap: group j+1 is in flight during the MFMAs
        // of group j (two named buffers, no copies: a copy would make the compiler wait for the load right away)
        auto load = [&](fbbev_v4f (&bfr)[NT], fbbev_v4f (&afr)[MT], int j) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                // unconditional load (a padded tap reads voxel 0, always mapped); the zero select happens at the point of
                // use: no divergent branch and no early consumer, so the loads of the next group overlap the MFMAs
                bfr[t] = *reinterpret_cast<const fbbev_v4f*>(x + off[t] + 16 * j);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                afr[mt] = *reinterpret_cast<const fbbev_v4f*>(wt + ((long long)j * mt_total + mt) * 256);
        };
        auto mma = [&](const fbbev_v4f (&braw)[NT], const fbbev_v4f (&afr)[MT], bool live) {
            fbbev_v4f bfr[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bfr[t] = (ok[t] && live) ? braw[t] : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_mfma_f32_16x16x4(afr[mt][e], bfr[t][e], acc[mt][t]);
        };
        fbbev_v4f b0[NT], a0[MT], b1[NT], a1[MT];
        load(b0, a0, 0);
        for (int j = 0; j < J; j += 2) {
            // straight-line body: both prefetches are unconditional (index clamped to the last group: a redundant, cached
            // re-read at the end of a tap) and for an odd group count the second MFMA block runs on a zero B operand
            // instead of being skipped.  Any branch here merges control-flow paths with different numbers of loads in
            // flight (the compiler then waits for all of them) or lets it sink a prefetch next to its consumer.
            // The scheduling fences pin the prefetch in front of the MFMA block it overlaps (the scheduler otherwise sinks
            // the loads behind the block to save registers and then waits for them at once).
            load(b1, a1, j + 1 < J ? j + 1 : J - 1);
            fbbev_sched_fence();
            mma(b0, a0, true);
            fbbev_sched_fence();
            load(b0, a0, j + 2 < J ? j + 2 : J - 1);
            fbbev_sched_fence();
            mma(b1, a1, j + 1 < J);
            fbbev_sched_fence();
        }
    }
    // epilogue: lane holds couts 16(mt0+mt) + 4g + {0..3} of voxel i of every tile
    const int pa = parity >> 2, pb = (parity >> 1) & 1, pc = parity & 1;
    const bool vec = (Cout & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!vq[t]) continue;
        long long ovox;
        if (mode == 1)
            ovox = (((long long)bq[t] * (2 * Do) + 2 * dq[t] + pa) * (2 * Ho) + 2 * hq[t] + pb) * (2 * Wo) + 2 * wq[t] + pc;
        else
            ovox = (((long long)bq[t] * Do + dq[t]) * Ho + hq[t]) * Wo + wq[t];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c0 = 16 * (mt0 + mt) + 4 * g;
            if (c0 >= Cout) continue;
            fbbev_v4f v = acc[mt][t] + *reinterpret_cast<const fbbev_v4f*>(bias + c0);     // bias is padded to 16*mt_total
            const long long o = ovox * Cout + c0;
            if (vec) {
                if (residual) v = v + *reinterpret_cast<const fbbev_v4f*>(residual + o);
                if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<fbbev_v4f*>(out + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (c0 + r >= Cout) break;
                    float s = v[r] + (residual ? residual[o + r] : 0.f);
                    out[o + r] = relu ? fmaxf(s, 0.f) : s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- soft-weighted multi-level blend of the head
// OccHead.forward_coarse_voxel (occupancy_head.py:159-170): every level is brought to the finest resolution with
// F.interpolate(trilinear, align_corners=False) and summed with per-voxel softmax weights -- three full-resolution
// 128-channel temporaries (328 MB each at 200x200x16) plus four multiply-add passes over them.  Here a lane owns 4
// channels of one fine voxel, takes level 0 directly, samples the (up to 3) coarse NDHWC maps itself (source index
// max(scale * (dst + 0.5) - 0.5, 0), scale = in / out, as ATen's area_pixel_compute_source_index; corners combined
// in ATen's nesting order) and writes the blended voxel once.  Bound: HBM (level 0 read + output written).
struct fbbev_blend_level {
    const float* f;          // (B, d, h, w, C)
    int d, h, w;
};

__device__ __forceinline__ void fbbev_lin_index(int dst, int in_size, int out_size, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)in_size / (float)out_size;
    float real = scale * ((float)dst + 0.5f) - 0.5f;
    real = real < 0.f ? 0.f : real;
    i0 = (int)real;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = real - (float)i0;
    l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256)
k_blend_levels_ndhwc(const float* __restrict__ x0, fbbev_blend_level lv1, fbbev_blend_level lv2, fbbev_blend_level lv3,
                     int n_coarse, const float* __restrict__ wsoft, int K, int B, int D, int H, int W, int C,
                     float* __restrict__ out) {
    const int quads = C >> 2;
    const long long total = (long long)B * D * H * W * quads;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % quads);
        long long v = idx / quads;
        const long long vox = v;
        const int wz = (int)(v % W); v /= W;
        const int hy = (int)(v % H); v /= H;
        const int dz = (int)(v % D);
        const int b = (int)(v / D);
        const float* ws = wsoft + vox * K;
        fbbev_v4f acc = *reinterpret_cast<const fbbev_v4f*>(x0 + vox * C + 4 * q) * ws[0];
        const fbbev_blend_level lv[3] = {lv1, lv2, lv3};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k >= n_coarse) break;
            int d0, d1, h0, h1, w0, w1;
            float ld0, ld1, lh0, lh1, lw0, lw1;
            fbbev_lin_index(dz, lv[k].d, D, d0, d1, ld0, ld1);
            fbbev_lin_index(hy, lv[k].h, H, h0, h1, lh0, lh1);
            fbbev_lin_index(wz, lv[k].w, W, w0, w1, lw0, lw1);
            const float* fb = lv[k].f + (long long)b * lv[k].d * lv[k].h * lv[k].w * C + 4 * q;
            auto at = [&](int d, int h, int w) {
                return *reinterpret_cast<const fbbev_v4f*>(fb + (((long long)d * lv[k].h + h) * lv[k].w + w) * C);
            };
            const fbbev_v4f t0 = (at(d0, h0, w0) * lw0 + at(d0, h0, w1) * lw1) * lh0 + (at(d0, h1, w0) * lw0 + at(d0, h1, w1) * lw1) * lh1;
            const fbbev_v4f t1 = (at(d1, h0, w0) * lw0 + at(d1, h0, w1) * lw1) * lh0 + (at(d1, h1, w0) * lw0 + at(d1, h1, w1) * lw1) * lh1;
            acc = acc + (t0 * ld0 + t1 * ld1) * ws[k + 1];
        }
        *reinterpret_cast<fbbev_v4f*>(out + vox * C + 4 * q) = acc;
    }
}

// ---------------------------------------------------------------- weight gradient (training)
// dW[tap][cout][cin] = sum over output voxels v of dY[v][cout] * X[src(v, tap)][cin]: per tap a GEMM with M = Cout,
// N = Cin and K = all output voxels.  A wave owns one (voxel chunk, 64 couts, 64 cins, tap) task: 16 accumulator tiles.
// K runs over voxels, whose rows are contiguous in channels, so here the M / N labelling is permuted instead of K:
// tile mt of the A operand holds couts {64mb + 4i + mt}, tile nt of the B operand cins {64nb + 4i + nt} (i = lane%16)
// -- a lane's ONE float4 of dY[v][64mb + 4i ..] is its A operand for the four M tiles and ONE float4 of
// X[src][64nb + 4i ..] its B operand for the four N tiles: 2 float4 loads (a voxel's 16 lanes read 256 contiguous bytes)
// feed 16 MFMAs, four k-steps (16 voxels) are batched per loop iteration with the next batch in flight (ping-pong).
// The voxel -> (b,d,h,w) decomposition is carried incrementally and branch-free (v advances by 4 per k-step).  Partial sums of the
// chunks meet in dW through fp32 atomic adds (dW is small; it must be zero on entry).  Requires Cout % 4 == Cin % 4 == 0.
template <int KS>
__global__ void __launch_bounds__(256)
k_conv3d_wgrad_ndhwc(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int B, int Di,
                     int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int stride, int pad, int chunk,
                     int n_chunks, int cout_blocks, int cin_blocks) {
    constexpr int T = KS * KS * KS, U = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kk = lane >> 4, i = lane & 15;
    long long task = (long long)blockIdx.x * 4 + wave;
    const long long n_tasks = (long long)n_chunks * cout_blocks * cin_blocks * T;
    if (task >= n_tasks) return;
    const int tap = (int)(task % T); task /= T;
    const int nb = (int)(task % cin_blocks); task /= cin_blocks;
    const int mb = (int)(task % cout_blocks);
    const int ch = (int)(task / cout_blocks);
    const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
    const long long nvox = (long long)B * Do * Ho * Wo;
    const long long v_begin = (long long)ch * chunk;
    const long long v_end = v_begin + chunk < nvox ? v_begin + chunk : nvox;
    const int ca = 64 * mb + 4 * i, cb = 64 * nb + 4 * i;
    const bool a_ok = ca < Cout, b_ok = cb < Cin;
    const int q4 = 4 / Wo, r4 = 4 % Wo;
    const int cac = a_ok ? ca : 0, cbc = b_ok ? cb : 0;
    // this lane's voxel of k-step 0 and its coordinates
    long long v = v_begin + kk;
    int wq, hq, dq, bq;
    {
        long long r = v < nvox ? v : 0;
        wq = (int)(r % Wo); r /= Wo;
        hq = (int)(r % Ho); r /= Ho;
        dq = (int)(r % Do);
        bq = (int)(r / Do);
    }
    fbbev_v4f acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    auto load = [&](fbbev_v4f (&af)[U], fbbev_v4f (&bf)[U], bool (&aok)[U], bool (&ok)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int di = dq * stride + kd - pad, hi = hq * stride + kh - pad, wi = wq * stride + kw - pad;
            const bool vin = v < v_end;
            const bool sin = vin & (di >= 0) & (di < Di) & (hi >= 0) & (hi < Hi) & (wi >= 0) & (wi < Wi);   // no short circuit: no branch
            // addresses from CLAMPED coordinates (always inside the tensors), validity applied by the selects below: a
            // conditional address computation would come out as a branch between the loads
            const long long vc = v < nvox ? v : nvox - 1;
            const int bqc = bq < B ? bq : B - 1;
            const int dic = di < 0 ? 0 : (di < Di ? di : Di - 1), hic = hi < 0 ? 0 : (hi < Hi ? hi : Hi - 1),
                      wic = wi < 0 ? 0 : (wi < Wi ? wi : Wi - 1);
            const long long ao = vc * Cout + cac;
            const long long bo = ((((long long)bqc * Di + dic) * Hi + hic) * Wi + wic) * Cin + cbc;
            af[u] = *reinterpret_cast<const fbbev_v4f*>(dy + ao);           // raw loads; the zero selects happen in mma()
            bf[u] = *reinterpret_cast<const fbbev_v4f*>(x + bo);
            aok[u] = vin & a_ok;
            ok[u] = sin & b_ok;
            // v += 4 in (b,d,h,w) digits without a data-dependent branch (a branch between the loads would also make the
            // compiler count them conservatively): w takes 4 % Wo with at most one wrap, the carries are at most 5
            v += 4;
            wq += r4;
            int c = wq >= Wo ? 1 : 0;
            wq -= c ? Wo : 0;
            hq += q4 + c;
#pragma unroll
            for (int rep = 0; rep < 5; ++rep) { c = hq >= Ho ? 1 : 0; hq -= c ? Ho : 0; dq += c; }
#pragma unroll
            for (int rep = 0; rep < 5; ++rep) { c = dq >= Do ? 1 : 0; dq -= c ? Do : 0; bq += c; }
        }
    };
    auto mma = [&](const fbbev_v4f (&araw)[U], const fbbev_v4f (&braw)[U], const bool (&aok)[U], const bool (&ok)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const fbbev_v4f bf = ok[u] ? braw[u] : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
            fbbev_v4f af[1];
            af[0] = aok[u] ? araw[u] : fbbev_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = fbbev_mfma_f32_16x16x4(af[0][mt], bf[nt], acc[mt][nt]);
        }
    };
    const int steps = (int)((v_end - v_begin + 4 * U - 1) / (4 * U));      // iterations of U k-steps (16 voxels)
    fbbev_v4f a0[U], b0[U], a1[U], b1[U];
    bool k0[U], k1[U], m0[U], m1[U];
    load(a0, b0, m0, k0);
    for (int s = 0; s < steps; s += 2) {
        load(a1, b1, m1, k1);                    // beyond the chunk every lane is masked: harmless, no branch
        fbbev_sched_fence();
        mma(a0, b0, m0, k0);
        fbbev_sched_fence();
        load(a0, b0, m0, k0);
        fbbev_sched_fence();
        mma(a1, b1, m1, k1);
        fbbev_sched_fence();
    }
    // D register r of a lane: row 4*kk + r -> cout 64mb + 4(4kk + r) + mt ; column i -> cin 64nb + 4i + nt
    float* dwt = dw + (long long)tap * Cout * Cin;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 64 * mb + 4 * (4 * kk + r) + mt;
            if (co >= Cout) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int ci = cb + nt;
                if (ci < Cin) fbbev_atomic_add_f32(dwt + (long long)co * Cin + ci, acc[mt][nt][r]);
            }
        }
}

// ---------------------------------------------------------------- bf16-MFMA variant of the forward convolution (inference)
// Same decomposition as k_conv3d_ndhwc -- a wave owns 64 voxels x 16*MT couts, activations and outputs stay fp32 NDHWC --
// but K is walked in 32-channel groups on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate): a lane's B operand is the 8
// consecutive channels 32j + 8(lane/16) .. +7 of its voxel (two float4 loads, rounded to bf16 in registers), its A operand
// 8 bf16 of host-prepared fragment-ordered weights (one 16-byte load)
//   wfb[parity][tap][j][mt][lane][e] = bf16( W[cout = 16mt + lane%16][cin = 32j + 8(lane/16) + e][tap] ).
// Both operands use the same slot -> channel rule, which is all the instruction needs (see rt.h).  Accumulation is fp32;
// results equal an fp32 convolution of the bf16-rounded inputs and weights up to summation order.  Cin % 32 == 0.
template <int KD, int KS, int MT>
__global__ void __launch_bounds__(256)
k_conv3d_ndhwc_bf16(const float* __restrict__ x, const unsigned short* __restrict__ wfb, const float* __restrict__ bias,
                    const float* __restrict__ residual, float* __restrict__ out, int B, int Di, int Hi, int Wi, int Cin,
                    int Do, int Ho, int Wo, int Cout, int mt_total, int stride, int pad, int relu, int mode,
                    long long wf_parity_stride, int gx, int gy) {
    constexpr int NT = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, i = lane & 15;
    const long long nvox = (long long)B * Do * Ho * Wo;
    const int bx = blockIdx.x % gx, by = (blockIdx.x / gx) % gy, parity = blockIdx.x / (gx * gy);
    const long long base = ((long long)bx * 4 + wave) * (16 * NT);
    if (base >= nvox) return;
    const int mt0 = by * MT;
    int bq[NT], dq[NT], hq[NT], wq[NT];
    bool vq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long n = base + 16 * t + i;
        vq[t] = n < nvox;
        long long r = vq[t] ? n : 0;
        wq[t] = (int)(r % Wo); r /= Wo;
        hq[t] = (int)(r % Ho); r /= Ho;
        dq[t] = (int)(r % Do);
        bq[t] = (int)(r / Do);
    }
    fbbev_v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    const int J = Cin >> 5;
    // fragment element = 8 bf16 = 16 bytes per lane: 64 lanes * 8 = 512 shorts per (tap, j, mt)
    const unsigned short* __restrict__ wfp = wfb + (long long)parity * wf_parity_stride + (long long)mt0 * 512 + lane * 8;
    const int pad_d = KD == 1 ? 0 : pad;
    for (int tap = 0; tap < KD * KS * KS; ++tap) {
        const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
        long long off[NT];
        bool ok[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int di = dq[t] * stride + kd - pad_d, hi = hq[t] * stride + kh - pad, wi = wq[t] * stride + kw - pad;
            ok[t] = vq[t] && di >= 0 && di < Di && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            off[t] = (ok[t] ? ((((long long)bq[t] * Di + di) * Hi + hi) * Wi + wi) * Cin : 0) + 8 * g;
        }
        const unsigned short* __restrict__ wt = wfp + (long long)tap * J * mt_total * 512;
        auto load = [&](fbbev_v4f (&blo)[NT], fbbev_v4f (&bhi)[NT], fbbev_bf16x8 (&afr)[MT], int j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                blo[t] = *reinterpret_cast<const fbbev_v4f*>(x + off[t] + 32 * j);
                bhi[t] = *reinterpret_cast<const fbbev_v4f*>(x + off[t] + 32 * j + 4);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) afr[mt] = fbbev_ld_bf16x8(wt + ((long long)j * mt_total + mt) * 512);
        };
        auto mma = [&](const fbbev_v4f (&blo)[NT], const fbbev_v4f (&bhi)[NT], const fbbev_bf16x8 (&afr)[MT], bool live) {
            fbbev_bf16x8 bfr[NT];
            const fbbev_v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool use = ok[t] && live;
                bfr[t] = fbbev_cvt_bf16x8(use ? blo[t] : zero, use ? bhi[t] : zero);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(afr[mt], bfr[t], acc[mt][t]);
        };
        fbbev_v4f l0[NT], h0[NT], l1[NT], h1[NT];
        fbbev_bf16x8 a0[MT], a1[MT];
        load(l0, h0, a0, 0);
        for (int j = 0; j < J; j += 2) {          // straight-line ping-pong body, as in k_conv3d_ndhwc
            load(l1, h1, a1, j + 1 < J ? j + 1 : J - 1);
            fbbev_sched_fence();
            mma(l0, h0, a0, true);
            fbbev_sched_fence();
            load(l0, h0, a0, j + 2 < J ? j + 2 : J - 1);
            fbbev_sched_fence();
            mma(l1, h1, a1, j + 1 < J);
            fbbev_sched_fence();
        }
    }
    const int pa = parity >> 2, pb = (parity >> 1) & 1, pc = parity & 1;
    const bool vec = (Cout & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!vq[t]) continue;
        long long ovox;
        if (mode == 1)
            ovox = (((long long)bq[t] * (2 * Do) + 2 * dq[t] + pa) * (2 * Ho) + 2 * hq[t] + pb) * (2 * Wo) + 2 * wq[t] + pc;
        else
            ovox = (((long long)bq[t] * Do + dq[t]) * Ho + hq[t]) * Wo + wq[t];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c0 = 16 * (mt0 + mt) + 4 * g;
            if (c0 >= Cout) continue;
            fbbev_v4f v = acc[mt][t] + *reinterpret_cast<const fbbev_v4f*>(bias + c0);
            const long long o = ovox * Cout + c0;
            if (vec) {
                if (residual) v = v + *reinterpret_cast<const fbbev_v4f*>(residual + o);
                if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<fbbev_v4f*>(out + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (c0 + r >= Cout) break;
                    float s = v[r] + (residual ? residual[o + r] : 0.f);
                    out[o + r] = relu ? fmaxf(s, 0.f) : s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- 3x3x3 stride-1 convolution, LDS-staged halo tile (bf16 MFMA)
// The direct kernels above re-read every input row 27 x (Cout/64) times from L2 / MALL; for the bf16 instruction that
// traffic, not the matrix pipe, is the bound.  Here a workgroup owns a compact 4 x 8 x 8 voxel tile (wave w = depth slice
// w, MFMA tile t = rows 2t, 2t+1 of that slice) and stages the tile's (4+2) x (8+2) x (8+2) = 600 halo rows ONCE per
// 32-channel group into LDS, already rounded to bf16 (64 bytes per row, 38.4 KB per buffer, two buffers): the 27 taps then
// read their B operands from LDS (16 bytes per lane: a voxel's 4 lanes read the row's 64 bytes; rows 4 apart share banks,
// i.e. the 4-way pattern a 1 KB wave read needs anyway) and only the A operands (weights, shared by all workgroups) come
// from L1 / L2.  Input rows are read 600 / 256 = 2.3 times per 64-channel output block instead of 27 times.
// Producer / consumer split: the workgroup is 8 waves -- waves 0..3 run the MFMAs, waves 4..7 only stage the next channel
// group into the other LDS buffer (one barrier per group).  The split is what makes the overlap real: a wave's memory
// counter retires loads in issue order, so a wave that both prefetched 20 row chunks and then needed its next weight
// fragment would have to wait for all of them; the loader waves have their own counters.  Zero padding and partial tiles
// are zero rows / skipped stores.
// Weights: the wfb layout of k_conv3d_ndhwc_bf16.  Requires Cin % 32 == 0; kernel 3, stride 1, padding 1 only.
template <int MT>
__global__ void __launch_bounds__(512)
k_conv3d_k3_tile_bf16(const float* __restrict__ x, const unsigned short* __restrict__ wfb, const float* __restrict__ bias,
                      const float* __restrict__ residual, float* __restrict__ out, int B, int D, int H, int W, int Cin,
                      int Cout, int mt_total, int relu, int tiles_d, int tiles_h, int tiles_w, int gy) {
    constexpr int TD = 4, TH = 8, TW = 8, HD = TD + 2, HH = TH + 2, HW = TW + 2, ROWS = HD * HH * HW;      // 600 halo rows
    constexpr int ITEMS = (ROWS * 4 + 255) / 256;                  // (row, 8-channel chunk) items per loader thread: 10
    unsigned short* lds = reinterpret_cast<unsigned short*>(fbbev_dyn_lds_f32());    // [2][ROWS][32] bf16
    const int tid = threadIdx.x, wave = (tid >> 6) & 3, lane = tid & 63;
    const bool loader = tid >= 256;
    const int g = lane >> 4, i = lane & 15;
    // XCD-aware order: workgroup n runs on XCD n % 8 (each XCD has its own L2).  The gy output-channel blocks of one voxel
    // tile are given consecutive slots of the SAME XCD, so the tile's input rows are fetched into that L2 once and its
    // other gy - 1 workgroups hit it; neighbouring tiles (shared halo) follow on the same XCD.
    const long long n_tiles = (long long)B * tiles_d * tiles_h * tiles_w;
    const long long slot = blockIdx.x >> 3;
    const int by = (int)(slot % gy);
    long long id = (slot / gy) * 8 + (blockIdx.x & 7);
    if (id >= n_tiles) return;                                       // whole workgroup, before any barrier
    const int tw = (int)(id % tiles_w); id /= tiles_w;
    const int th = (int)(id % tiles_h); id /= tiles_h;
    const int td = (int)(id % tiles_d);
    const int b = (int)(id / tiles_d);
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int mt0 = by * MT;
    const int J = Cin >> 5;
    auto stage = [&](int j, int buf) {                               // loader waves: group j's halo rows -> LDS buffer `buf`
        const fbbev_v4f zero = {0.f, 0.f, 0.f, 0.f};
        fbbev_v4f slo[ITEMS], shi[ITEMS];
        bool in[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const int item = (tid - 256) + 256 * q;
            const int row = item >> 2, chunk = item & 3;
            const int rd = row / (HH * HW), rh = (row / HW) % HH, rw = row % HW;
            const int vd = d0 - 1 + rd, vh = h0 - 1 + rh, vw = w0 - 1 + rw;
            in[q] = item < ROWS * 4 && vd >= 0 && vd < D && vh >= 0 && vh < H && vw >= 0 && vw < W;
            const float* p = x + (in[q] ? ((((long long)b * D + vd) * H + vh) * W + vw) * Cin + 8 * chunk + 32 * j : 0);
            slo[q] = *reinterpret_cast<const fbbev_v4f*>(p);         // all loads first, then the conversions and LDS stores
            shi[q] = *reinterpret_cast<const fbbev_v4f*>(p + 4);
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const int item = (tid - 256) + 256 * q;
            if (item >= ROWS * 4) continue;
            *reinterpret_cast<fbbev_bf16x8*>(lds + buf * (ROWS * 32) + (item >> 2) * 32 + 8 * (item & 3)) =
                fbbev_cvt_bf16x8(in[q] ? slo[q] : zero, in[q] ? shi[q] : zero);
        }
    };
    // compute waves: this lane's voxel of each MFMA tile -- depth slice = wave, rows 2t + i/8, column i%8 -> halo row of tap (0,0,0)
    int hrow[4];
    bool vq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int lh = 2 * t + (i >> 3), lw = i & 7;
        hrow[t] = (wave * HH + lh) * HW + lw;
        vq[t] = !loader && d0 + wave < D && h0 + lh < H && w0 + lw < W;
    }
    fbbev_v4f acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
    const unsigned short* __restrict__ wfp = wfb + (long long)mt0 * 512 + lane * 8;
    if (loader) stage(0, 0);
    __syncthreads();
    for (int j = 0; j < J; ++j) {
        const int buf = j & 1;
        if (loader) {
            if (j + 1 < J) stage(j + 1, buf ^ 1);                    // nobody reads that buffer during this iteration
        } else {
            const unsigned short* lb = lds + buf * (ROWS * 32) + 8 * g;
            fbbev_bf16x8 a0[MT], a1[MT];
            auto loadA = [&](fbbev_bf16x8 (&afr)[MT], int tap) {
                const unsigned short* wt = wfp + ((long long)tap * J + j) * mt_total * 512;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) afr[mt] = fbbev_ld_bf16x8(wt + (long long)mt * 512);
            };
            auto mma = [&](const fbbev_bf16x8 (&afr)[MT], int tap, bool live) {
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int shift = (kd * HH + kh) * HW + kw;
                const fbbev_bf16x8 zero = fbbev_cvt_bf16x8(fbbev_v4f{0.f, 0.f, 0.f, 0.f}, fbbev_v4f{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const fbbev_bf16x8 raw = fbbev_ld_bf16x8(lb + (hrow[t] + shift) * 32);
                    const fbbev_bf16x8 bfr = live ? raw : zero;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][t] = fbbev_mfma_f32_16x16x32_bf16(afr[mt], bfr, acc[mt][t]);
                }
            };
            // weights ping-pong across the 27 taps: straight-line body with clamped prefetch indices and scheduling fences
            // (see k_conv3d_ndhwc); the 28th slot runs on a zero B operand
            loadA(a0, 0);
            for (int tap = 0; tap < 27; tap += 2) {
                loadA(a1, tap + 1 < 27 ? tap + 1 : 26);
                fbbev_sched_fence();
                mma(a0, tap, true);
                fbbev_sched_fence();
                loadA(a0, tap + 2 < 27 ? tap + 2 : 26);
                fbbev_sched_fence();
                mma(a1, tap + 1 < 27 ? tap + 1 : 26, tap + 1 < 27);
                fbbev_sched_fence();
            }
        }
        __syncthreads();
    }
    // epilogue (as k_conv3d_ndhwc): lane holds couts 16(mt0+mt) + 4g + {0..3} of voxel i of every tile
    const bool vec = (Cout & 3) == 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (!vq[t]) continue;
        const int lh = 2 * t + (i >> 3), lw = i & 7;
        const long long ovox = (((long long)b * D + d0 + wave) * H + h0 + lh) * W + w0 + lw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c0 = 16 * (mt0 + mt) + 4 * g;
            if (c0 >= Cout) continue;
            fbbev_v4f v = acc[mt][t] + *reinterpret_cast<const fbbev_v4f*>(bias + c0);
            const long long o = ovox * Cout + c0;
            if (vec) {
                if (residual) v = v + *reinterpret_cast<const fbbev_v4f*>(residual + o);
                if (relu) v = fbbev_v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<fbbev_v4f*>(out + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (c0 + r >= Cout) break;
                    float s = v[r] + (residual ? residual[o + r] : 0.f);
                    out[o + r] = relu ? fmaxf(s, 0.f) : s;
                }
            }
        }
    }
}
