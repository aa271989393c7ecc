// history_kernels.h -- temporal history alignment of FB-OCC.
//
// Replaces FBOCC.generate_grid + F.grid_sample of FBOCC.fuse_history
// (mmdet3d/models/fbbev/detectors/fbocc.py:169-205, 264-275): the reference materialises a (B,Y,X,Z,4,1) homogeneous
// grid, multiplies it by a batched 4x4 (rt_flow), normalises, permutes, and hands a (B,Z,Y,X,3) grid to the
// generic 5-D grid_sample -- then cats / clones the 16-frame history (410 MB per sample at 100x100x8, fp32)
// three more times.  Here
//   k_history_flow : rt_flow[b] = inv(feat2bev) . history_forward_augs[b] . curr_to_prev_ego_rt[b]
//                    . inv(forward_augs[b]) . feat2bev   (fbocc.py:184-203), one thread per sample, closed-form
//                    inverses (feat2bev is scale+translation, forward_augs is the bda block), products
//                    associated left to right like the reference expression.
//   k_history_warp : a thread owns an output voxel (x fastest => coalesced taps for near-translations), evaluates
//                    rt_flow . (x,y,z,1), repeats the reference's normalise / un-normalise arithmetic
//                    (align_corners=True), builds the 8 trilinear taps once (zero padding: weight 0) and then
//                    streams over a group of channels: 8 gathers + 8 fma + 1 store per channel.  The sampling grid
//                    is never materialised and the output can be written straight into the frame slots of the
//                    next history buffer (out_stride_b).
// Bound: HBM (history read once through L2 + written once).
#pragma once
#include "rt.h"
#include "geom_kernels.h"
#include "pool_kernels.h"      // fbbev_f32_to_f16: the integer-only binary32 -> binary16 rounding (same bits on the emulator)

// Storage element of the history ring: ET 0 = f32, 1 = bf16, 2 = f16 (BASELINE configs[4] names fp16).  All arithmetic is
// fp32: elements are widened exactly at the load and rounded ONCE (nearest-even) at the store.
template <int ET>
__device__ __forceinline__ float fbbev_ld_elem(const void* base, long long i) {
    if constexpr (ET == 0) return static_cast<const float*>(base)[i];
    else {
        const unsigned int h = static_cast<const unsigned short*>(base)[i];
        if constexpr (ET == 1) { const unsigned int u = h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
        else return fbbev_f16_bits_to_f32(h);        // rt.h: v_cvt_f32_f16 on the GPU (exact for every half, subnormals included)
    }
}
// raw element (no conversion: a prefetch must not wait for its own load) and its exact widening at the point of use
template <int ET>
__device__ __forceinline__ unsigned int fbbev_ld_raw(const void* base, long long i) {
    if constexpr (ET == 0) return static_cast<const unsigned int*>(base)[i];
    else return static_cast<const unsigned short*>(base)[i];
}
template <int ET>
__device__ __forceinline__ float fbbev_widen(unsigned int r) {
    if constexpr (ET == 2) return fbbev_f16_bits_to_f32(r);
    else {
        const unsigned int u = (ET == 1) ? (r << 16) : r;
        float f;
        __builtin_memcpy(&f, &u, 4);
        return f;
    }
}
template <int ET>
__device__ __forceinline__ void fbbev_st_elem(void* base, long long i, float v) {
    if constexpr (ET == 0) static_cast<float*>(base)[i] = v;
    else static_cast<unsigned short*>(base)[i] = (unsigned short)(fbbev_pack2<ET>(v, 0.f) & 0xffffu);
}

__device__ __forceinline__ void fbbev_mat4(const float* a, const float* b, float* o) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            o[r * 4 + c] = a[r * 4] * b[c] + a[r * 4 + 1] * b[4 + c] + a[r * 4 + 2] * b[8 + c] + a[r * 4 + 3] * b[12 + c];
}

// dx3 = voxel size, t3 = feat2bev translation (bx - dx/2 = grid lower bound), both (x,y,z)
__global__ void __launch_bounds__(64)
k_history_flow(const float* __restrict__ hist_augs, const float* __restrict__ ego, const float* __restrict__ bda,
               float dx0, float dx1, float dx2, float t0, float t1, float t2, int B, float* __restrict__ flow) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float f2b[16] = {dx0, 0.f, 0.f, t0, 0.f, dx1, 0.f, t1, 0.f, 0.f, dx2, t2, 0.f, 0.f, 0.f, 1.f};
    float inv[16] = {1.f / dx0, 0.f, 0.f, -t0 / dx0, 0.f, 1.f / dx1, 0.f, -t1 / dx1, 0.f, 0.f, 1.f / dx2, -t2 / dx2,
                     0.f, 0.f, 0.f, 1.f};
    float ib[9], fi[16], a[16], c[16];
    fbbev_inv3(bda + b * 9, ib);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) fi[r * 4 + k] = ib[r * 3 + k];
        fi[r * 4 + 3] = 0.f;
    }
    fi[12] = fi[13] = fi[14] = 0.f; fi[15] = 1.f;
    fbbev_mat4(inv, hist_augs + b * 16, a);
    fbbev_mat4(a, ego + b * 16, c);
    fbbev_mat4(c, fi, a);
    fbbev_mat4(a, f2b, c);
    for (int k = 0; k < 16; ++k) flow[b * 16 + k] = c[k];
}

// work item = ((b * n_groups) + group) * n_chunks + chunk
// The 8 trilinear taps of output voxel (x, y, z) under the sample's rt_flow `m` (4x4 row-major): voxel index of each tap in
// the source frame (0 for a tap outside the grid) and its weight (0 outside).  fbocc.py:205 rt_flow @ (x,y,z,1); :208-209
// normalise; ATen grid_sampler_unnormalize (align_corners=True) -- the reference's normalise / un-normalise round trip is
// repeated so that the coordinates carry the same rounding.  Every operation is ONE correctly rounded fp32 operation
// (fbbev_mul / fbbev_add / fbbev_sub / fbbev_div of rt.h: built without the `contract` flag): the planar kernel, the voxel-major kernel and the fused warp-and-convolution kernel call
// this one function and therefore blend the same taps with the same weights whatever code surrounds the call -- with plain
// `a * b + c` the compiler contracts differently from kernel to kernel and the rings differed in the last bit.
// Tap order tnw,tne,tsw,tse,bnw,bne,bsw,bse: x fastest, then y, then z.  NaN / huge coordinates fail every bounds test =>
// all weights 0 => output 0 (zero padding).
__device__ __forceinline__ void fbbev_warp_taps(const float* __restrict__ m, int x, int y, int z, int X, int Y, int Z,
                                                int (&tv)[8], float (&w)[8]) {
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    float gx = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[0], fx), fbbev_mul(m[1], fy)), fbbev_mul(m[2], fz)), m[3]);
    float gy = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[4], fx), fbbev_mul(m[5], fy)), fbbev_mul(m[6], fz)), m[7]);
    float gz = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[8], fx), fbbev_mul(m[9], fy)), fbbev_mul(m[10], fz)), m[11]);
    gx = fbbev_sub(fbbev_mul(fbbev_div(gx, (float)(X - 1)), 2.0f), 1.0f);
    gy = fbbev_sub(fbbev_mul(fbbev_div(gy, (float)(Y - 1)), 2.0f), 1.0f);
    gz = fbbev_sub(fbbev_mul(fbbev_div(gz, (float)(Z - 1)), 2.0f), 1.0f);
    const float ix = fbbev_mul(fbbev_div(fbbev_add(gx, 1.f), 2.f), (float)(X - 1));
    const float iy = fbbev_mul(fbbev_div(fbbev_add(gy, 1.f), 2.f), (float)(Y - 1));
    const float iz = fbbev_mul(fbbev_div(fbbev_add(gz, 1.f), 2.f), (float)(Z - 1));
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const float wx1 = fbbev_sub(ix, x0f), wy1 = fbbev_sub(iy, y0f), wz1 = fbbev_sub(iz, z0f);
    const float wx0 = fbbev_sub(fbbev_add(x0f, 1.f), ix), wy0 = fbbev_sub(fbbev_add(y0f, 1.f), iy), wz0 = fbbev_sub(fbbev_add(z0f, 1.f), iz);
    const bool fin = (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f) && (fabsf(iz) < 1.0e9f);
    const int x0 = fin ? (int)x0f : -2, y0 = fin ? (int)y0f : -2, z0 = fin ? (int)z0f : -2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cx = x0 + (k & 1), cy = y0 + ((k >> 1) & 1), cz = z0 + (k >> 2);
        const bool ok = cx >= 0 && cx < X && cy >= 0 && cy < Y && cz >= 0 && cz < Z;
        const float wk = fbbev_mul(fbbev_mul((k & 1) ? wx1 : wx0, ((k >> 1) & 1) ? wy1 : wy0), (k >> 2) ? wz1 : wz0);
        tv[k] = ok ? (cz * Y + cy) * X + cx : 0;
        w[k] = ok ? wk : 0.f;
    }
}

template <int ET>
__global__ void __launch_bounds__(256)
k_history_warp(const void* __restrict__ hist, long long hist_stride_b, const float* __restrict__ flow, int CH,
               int Z, int Y, int X, int ch_per_block, int n_groups, int n_chunks, int per_xcd, int n_work,
               void* __restrict__ out, long long out_stride_b) {
    // workgroup b runs on XCD b%8: give each XCD one contiguous eighth of the (sample, channel group, chunk) space so
    // that the y-neighbour taps of adjacent chunks are re-used from that XCD's own L2
    const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= n_work) return;
    const int chunk = work % n_chunks;
    const int bg = work / n_chunks;
    const int grp = bg % n_groups, b = bg / n_groups;
    const int YX = Y * X, ZYX = Z * YX;
    const int v = chunk * 256 + threadIdx.x;
    if (v >= ZYX) return;
    const int z = v / YX, r = v - z * YX, y = r / X, x = r - y * X;
    int off[8];
    float w[8];
    fbbev_warp_taps(flow + b * 16, x, y, z, X, Y, Z, off, w);
    const int c0 = grp * ch_per_block;
    const int c1 = (c0 + ch_per_block < CH) ? c0 + ch_per_block : CH;
    // element offsets of this workgroup's first channel (16-bit storage: the same offsets, half the bytes)
    long long so = (long long)b * hist_stride_b + (long long)c0 * ZYX;
    long long dof = (long long)b * out_stride_b + (long long)c0 * ZYX + v;
    int c = c0;
    constexpr int U = 4;                           // channels in flight per thread: 8*U independent gathers
    for (; c + U <= c1; c += U) {
        float a[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[u][k] = fbbev_ld_elem<ET>(hist, so + (long long)u * ZYX + off[k]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s0 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s0 = fmaf(a[u][k], w[k], s0);
            fbbev_st_elem<ET>(out, dof + (long long)u * ZYX, s0);
        }
        so += U * (long long)ZYX;
        dof += U * (long long)ZYX;
    }
    for (; c < c1; ++c) {
        float s0 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s0 = fmaf(fbbev_ld_elem<ET>(hist, so + off[k]), w[k], s0);
        fbbev_st_elem<ET>(out, dof, s0);
        so += ZYX;
        dof += ZYX;
    }
}

// ---------------------------------------------------------------- voxel-major ring (opt-in: ring_layout = voxel_major)
// The reference keeps the history as (B, T*C, Z, Y, X): 80 channel planes per frame, 5 MB apart at 400x400x16.  On that
// layout every tap of k_history_warp is a 2- or 4-byte gather and the vector L1's access rate is the bound (0.2-0.3 of the
// HBM peak).  In the voxel-major ring a frame is [voxel][channel] (x fastest over voxels): the C elements of a voxel are
// contiguous (160 bytes at C = 80 in 16 bits), a tap is a 16-byte load of VE = 8 (16-bit) or 4 (fp32) channels, the taps
// of x-neighbours are adjacent, and the fused convolutions read their MFMA operands as rows.  A thread owns (voxel, channel
// group of VE) and TU consecutive frames: 8 TU independent 16-byte loads in flight, 8 fmaf per element in k_history_warp's
// tap order with k_history_warp's weights -- the elements are the SAME bits as the planar kernel's (tested), only
// their addresses differ.
template <int ET>
__device__ __forceinline__ void fbbev_widen_vec(fbbev_v4u raw, float* f) {      // 16 bytes -> VE floats, exact
    if constexpr (ET == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = fbbev_widen<0>(raw[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = fbbev_widen<ET>(raw[e] & 0xffffu);
            f[2 * e + 1] = fbbev_widen<ET>(raw[e] >> 16);
        }
    }
}
template <int ET>
__device__ __forceinline__ fbbev_v4u fbbev_narrow_vec(const float* f) {          // VE floats -> 16 bytes, one nearest-even rounding
    fbbev_v4u r;
    if constexpr (ET == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsigned int u; __builtin_memcpy(&u, &f[e], 4); r[e] = u; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fbbev_cvt_pk16<ET>(f[2 * e], f[2 * e + 1]);      // rt.h: the conversion instructions
    }
    return r;
}

// Source taps of output voxel (x, y, z) under the sample's rt_flow `m` (4x4 row-major) on a voxel-major frame: byte offset
// of the channel group starting at channel c0 in each of the 8 trilinear taps (0 for a tap outside the grid) and the tap
// weights (0 outside).  EXACTLY the expression sequence of k_history_warp (the reference's normalise / un-normalise round
// trip, fbocc.py:197-203,275), shared by k_history_warp_vm and the fused warp-and-convolution kernel: same taps, same weights.
__device__ __forceinline__ void fbbev_warp_taps_vm(const float* __restrict__ m, int x, int y, int z, int X, int Y, int Z, int C,
                                                   int c0, int esz, unsigned int (&ob)[8], float (&w)[8]) {
    int tv[8];
    fbbev_warp_taps(m, x, y, z, X, Y, Z, tv, w);
#pragma unroll
    for (int k = 0; k < 8; ++k) ob[k] = ((unsigned int)tv[k] * (unsigned int)C + (unsigned int)c0) * (unsigned int)esz;
}

// A workgroup = 256 (voxel, channel group) pairs of ONE grid row (x, group fastest: 25.6 voxels at C = 80 in 16 bits); a
// thread walks the T frames of its sample TU at a time (the flow, hence the taps, is the sample's: set up once): uniform
// frame base + 32-bit lane byte offsets, 8 TU loads in flight.
// Workgroup order: z fastest, then y inside a band of YB rows, then the x chunk, then the band, then the sample -- the ~256
// workgroups an XCD runs at a time form a (z, y) slab, so the y+1 and z+1 taps of one workgroup are the x-row of another
// one in flight on the same L2 (in flattened voxel order the z neighbour is 6250 workgroups away at 400x400 and every tap
// plane came from HBM again: 2.9x the algorithmic reads, profiles/r02_pmc_history_vm.json).
template <int ET, int TU, int ST>
__global__ void __launch_bounds__(256)
k_history_warp_vm(const void* __restrict__ hist, long long hist_stride_b, const float* __restrict__ flow, int T, int C,
                  int Z, int Y, int X, int groups, int n_xc, int YB, int nyb, int per_xcd, int n_work,
                  void* __restrict__ out, long long out_stride_b, int yb0) {
    constexpr int VE = ET == 0 ? 4 : 8;
    constexpr int ESZ = ET == 0 ? 4 : 2;
    int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);            // one contiguous eighth per XCD
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= n_work) return;
    const int z = work % Z; work /= Z;
    const int yl = work % YB; work /= YB;
    const int xc = work % n_xc; work /= n_xc;
    const int yb = yb0 + work % nyb, b = work / nyb;    // the launch covers the bands yb0 .. yb0 + nyb - 1 (all of them, or one chunk of the pipelined step)
    const int y = yb * YB + yl;
    const int item = xc * 256 + (int)threadIdx.x;
    const int x = item / groups, gq = item - x * groups;
    if (y >= Y || x >= X) return;
    const int YX = Y * X, ZYX = Z * YX;
    const int v = (z * Y + y) * X + x;
    unsigned int ob[8];                                 // byte offset of the tap's channel group inside a frame (< 4 GiB: checked)
    float w[8];
    fbbev_warp_taps_vm(flow + b * 16, x, y, z, X, Y, Z, C, gq * VE, ESZ, ob, w);
    const size_t frame_bytes = (size_t)ZYX * C * ESZ;
    const char* src = static_cast<const char*>(hist) + (size_t)b * hist_stride_b * ESZ;
    char* dst = static_cast<char*>(out) + (size_t)b * out_stride_b * ESZ + ((size_t)v * C + gq * VE) * ESZ;
    for (int t0 = 0; t0 < T; t0 += TU) {
        // straight-line body: the frame index is clamped for the loads AND the stores (an odd tail computes and stores frame
        // T-1 twice, the same bits) -- with a break in the store loop the compiler split the batch into one frame at a time
        fbbev_v4u raw[TU][8];
        size_t fo[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            fo[u] = (size_t)(t0 + u < T ? t0 + u : T - 1) * frame_bytes;   // uniform
#pragma unroll
            for (int k = 0; k < 8; ++k) raw[u][k] = *reinterpret_cast<const fbbev_v4u*>(src + fo[u] + ob[k]);
        }
        fbbev_sched_fence();
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            float acc[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (ET == 2) {                 // the half is widened by the multiply itself (same value, half the issue slots)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] = fbbev_fma_f16<0>(raw[u][k][e], w[k], acc[2 * e]);
                        acc[2 * e + 1] = fbbev_fma_f16<1>(raw[u][k][e], w[k], acc[2 * e + 1]);
                    }
                } else {
                    float a[VE];
                    fbbev_widen_vec<ET>(raw[u][k], a);
#pragma unroll
                    for (int e = 0; e < VE; ++e) acc[e] = fmaf(a[e], w[k], acc[e]);
                }
            }
            // write-once stream: non-temporal, so the new frames do not push the tap rows of the neighbouring workgroups out of L2
            const fbbev_v4u pk = fbbev_narrow_vec<ET>(acc);
            fbbev_v4f pf;
            __builtin_memcpy(&pf, &pk, 16);
            fbbev_store4<ST>(reinterpret_cast<float*>(dst + fo[u]), pf);
        }
    }
}

// current frame (B, C, N) fp32 planes -> slot [b][N][C] of a voxel-major ring (one nearest-even rounding for a 16-bit ring).
// A workgroup turns 64 plane positions: coalesced 256-byte plane reads into an LDS tile [C][65], 16-byte row writes out
// of it.  inner > 1: the planes are stored (Y, X, Z) like the BEV volume the view transformation hands over (fbocc.py:212
// permutes it for grid_sample) -- plane position i = (y*X + x)*Z + z lands in row n = z*(Y*X) + (y*X + x); inner = Z.
template <int ET>
__global__ void __launch_bounds__(256)
k_history_frame_vm(const float* __restrict__ curr, int C, int N, int inner, int tiles_per_b, void* __restrict__ out,
                   long long out_stride_b) {
    constexpr int VE = ET == 0 ? 4 : 8;
    float* tile = fbbev_dyn_lds_f32();
    const int b = blockIdx.x / tiles_per_b, n0 = (blockIdx.x - b * tiles_per_b) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cb = curr + (long long)b * C * N;
    // round 6: a wave's plane pieces are REQUESTED ten at a time before the first is stored (as one load + LDS store per iteration the
    // loop was 20 dependent round trips per wave: 0.33 ms for 1.2 GB at 400x400x16)
    {
        constexpr int U = 10;
        const long long at = (n0 + lane < N) ? n0 + lane : 0;
        for (int c0 = wave; c0 < C; c0 += 4 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int c = c0 + 4 * u; v[u] = cb[(long long)(c < C ? c : c0) * N + at]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { const int c = c0 + 4 * u; if (c < C) tile[c * 65 + lane] = (n0 + lane < N) ? v[u] : 0.f; }
        }
    }
    __syncthreads();
    const int groups = C / VE, plane = N / inner;
    fbbev_v4u* dst = reinterpret_cast<fbbev_v4u*>(out);
    for (int i = threadIdx.x; i < 64 * groups; i += 256) {
        const int vl = i / groups, gq = i - vl * groups;
        const int src = n0 + vl;
        if (src >= N) continue;
        const int row = inner > 1 ? (src % inner) * plane + src / inner : src;
        float f[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) f[e] = tile[(gq * VE + e) * 65 + vl];
        dst[((long long)b * out_stride_b + (long long)row * C + gq * VE) / VE] = fbbev_narrow_vec<ET>(f);
    }
}

// ---------------------------------------------------------------- LDS-staged variant (opt-in: FBBEV_HISTORY_WARP=lds)
// STATUS (round 2, measured): bit-identical to k_history_warp on the GPU, but 3x slower as built (REF 1.05 vs 0.36 ms,
// 400x400x16 25.6 vs 10.7 ms): with one 1024-thread workgroup per CU the staging loads, the barrier and the taps of a
// channel run back to back with nothing to overlap them.  What it needs next: 256-thread bricks (4 workgroups per CU) or a
// register-prefetched double buffer, and several channels per staging round.  Not the default.
// k_history_warp above is bound by the vector L1's access rate (8 dword gathers per output: 0.30 of the HBM peak; with a
// 16-bit ring the same gathers move half the bytes: 0.20): for the near-rigid flows of ego motion the 8 taps of
// neighbouring voxels overlap 8-fold.  Here a 1024-thread workgroup owns an output brick of BZ x TY x TX <= 4096 voxels
// (x fastest: a wave stores 64 consecutive x), computes the bounding box of the brick's source coordinates once (the flow
// is affine: the extremes are at the 8 corners; +-1 voxel of slack covers the second tap and the fp32 rounding of the
// chain), and per channel stages that box into LDS with coalesced row loads -- every source element is fetched once per
// brick instead of ~8 times -- then takes the 8 taps from LDS.  Tap order, weights and the fmaf chain are those of
// k_history_warp: the two kernels produce the same bits (tested).  A brick whose box does not fit (large rotations)
// gathers from global memory as before.
#define FBBEV_HW_BOX_FLOATS 12288                 // 48 KiB of LDS per workgroup: two workgroups per CU
#define FBBEV_HW_VPT 4                            // voxels per thread (brick <= 4096 voxels, 1024 threads)
#define FBBEV_HW_PITCH 80                         // LDS row pitch of the staged box (x extent <= 80: TX = 64 + halo)

template <int ET>
__global__ void __launch_bounds__(1024)
k_history_warp_lds(const void* __restrict__ hist, long long hist_stride_b, const float* __restrict__ flow, int CH,
                   int Z, int Y, int X, int BZ, int TY, int TX, int ntz, int nty, int ntx, int ch_per_block,
                   int n_groups, void* __restrict__ out, long long out_stride_b) {
    __shared__ float box[FBBEV_HW_BOX_FLOATS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int work = blockIdx.x;
    const int tx = work % ntx; work /= ntx;
    const int ty = work % nty; work /= nty;
    const int tz = work % ntz; work /= ntz;
    const int grp = work % n_groups, b = work / n_groups;
    const int X0 = tx * TX, Y0 = ty * TY, Z0 = tz * BZ;
    const int YX = Y * X, ZYX = Z * YX;
    const float* m = flow + b * 16;
    // source coordinate of an output voxel: EXACTLY the expression sequence of k_history_warp
    auto src = [&](float fx, float fy, float fz, float& ix, float& iy, float& iz) {
        // single correctly rounded operations, as fbbev_warp_taps (no contraction: the same coordinates in every kernel)
        float gx = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[0], fx), fbbev_mul(m[1], fy)), fbbev_mul(m[2], fz)), m[3]);
        float gy = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[4], fx), fbbev_mul(m[5], fy)), fbbev_mul(m[6], fz)), m[7]);
        float gz = fbbev_add(fbbev_add(fbbev_add(fbbev_mul(m[8], fx), fbbev_mul(m[9], fy)), fbbev_mul(m[10], fz)), m[11]);
        gx = fbbev_sub(fbbev_mul(fbbev_div(gx, (float)(X - 1)), 2.0f), 1.0f);
        gy = fbbev_sub(fbbev_mul(fbbev_div(gy, (float)(Y - 1)), 2.0f), 1.0f);
        gz = fbbev_sub(fbbev_mul(fbbev_div(gz, (float)(Z - 1)), 2.0f), 1.0f);
        ix = fbbev_mul(fbbev_div(fbbev_add(gx, 1.f), 2.f), (float)(X - 1));
        iy = fbbev_mul(fbbev_div(fbbev_add(gy, 1.f), 2.f), (float)(Y - 1));
        iz = fbbev_mul(fbbev_div(fbbev_add(gz, 1.f), 2.f), (float)(Z - 1));
    };
    // bounding box of the brick's sources (uniform: every thread evaluates the 8 corners)
    const int X1 = (X0 + TX < X ? X0 + TX : X) - 1, Y1 = (Y0 + TY < Y ? Y0 + TY : Y) - 1, Z1 = (Z0 + BZ < Z ? Z0 + BZ : Z) - 1;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    bool finite = true;
    for (int k = 0; k < 8; ++k) {
        float c[3];
        src((float)((k & 1) ? X1 : X0), (float)((k & 2) ? Y1 : Y0), (float)((k & 4) ? Z1 : Z0), c[0], c[1], c[2]);
        for (int a = 0; a < 3; ++a) {
            finite = finite && (fabsf(c[a]) < 1.0e9f);
            lo[a] = fminf(lo[a], c[a]);
            hi[a] = fmaxf(hi[a], c[a]);
        }
    }
    int b0[3] = {0, 0, 0}, bs[3] = {0, 0, 0};
    const int dim[3] = {X, Y, Z};
    bool use_lds = finite;
    if (finite) {
        for (int a = 0; a < 3; ++a) {
            int l = (int)floorf(lo[a]) - 1, h = (int)floorf(hi[a]) + 2;      // taps floor and floor + 1, one voxel of slack
            l = l < 0 ? 0 : l;
            h = h > dim[a] - 1 ? dim[a] - 1 : h;
            b0[a] = l;
            bs[a] = h >= l ? h - l + 1 : 0;                                  // 0: the whole brick samples outside the grid
        }
        use_lds = bs[0] <= FBBEV_HW_PITCH && (long long)FBBEV_HW_PITCH * bs[1] * bs[2] <= FBBEV_HW_BOX_FLOATS;
    }
    const int bxs = bs[0], bys = bs[1], bzs = bs[2];
    const bool empty = use_lds && (bxs == 0 || bys == 0 || bzs == 0);
    // this thread's voxels: brick-local id j = tid + v * 1024, x fastest.  Per tap: weight (0 for a tap outside the grid)
    // and its LDS offset packed 2 x 16 bit (0 for an invalid tap: "0 x a staged element", like the gather kernel's
    // offset 0) -- no per-tap predicate survives into the channel loop (the compiler would keep 64 lane masks alive).
    int vox[FBBEV_HW_VPT];
    float w[FBBEV_HW_VPT][8];
    unsigned int loff[FBBEV_HW_VPT][4];
    bool miss = false;                               // a valid tap outside the staged box (slack exceeded)
    const int nbrick = BZ * TY * TX;
#pragma unroll
    for (int v = 0; v < FBBEV_HW_VPT; ++v) {
        const int j = tid + v * 1024;
        vox[v] = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[v][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) loff[v][k] = 0u;
        if (j >= nbrick) continue;
        const int lx = j % TX, ly = (j / TX) % TY, lz = j / (TX * TY);
        const int x = X0 + lx, y = Y0 + ly, z = Z0 + lz;
        if (x >= X || y >= Y || z >= Z) continue;
        vox[v] = (z * Y + y) * X + x;
        float ix, iy, iz;
        src((float)x, (float)y, (float)z, ix, iy, iz);
        const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
        const float wx1 = fbbev_sub(ix, x0f), wy1 = fbbev_sub(iy, y0f), wz1 = fbbev_sub(iz, z0f);
        const float wx0 = fbbev_sub(fbbev_add(x0f, 1.f), ix), wy0 = fbbev_sub(fbbev_add(y0f, 1.f), iy), wz0 = fbbev_sub(fbbev_add(z0f, 1.f), iz);
        const bool fin = (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f) && (fabsf(iz) < 1.0e9f);
        const int x0 = fin ? (int)x0f : -2, y0 = fin ? (int)y0f : -2, z0 = fin ? (int)z0f : -2;
        const int base = ((z0 - b0[2]) * bys + (y0 - b0[1])) * FBBEV_HW_PITCH + (x0 - b0[0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {                 // order tnw,tne,tsw,tse,bnw,bne,bsw,bse: x fastest, then y, then z
            const int cx = x0 + (k & 1), cy = y0 + ((k >> 1) & 1), cz = z0 + (k >> 2);
            const bool ok = cx >= 0 && cx < X && cy >= 0 && cy < Y && cz >= 0 && cz < Z;
            const float wk = fbbev_mul(fbbev_mul((k & 1) ? wx1 : wx0, ((k >> 1) & 1) ? wy1 : wy0), (k >> 2) ? wz1 : wz0);
            w[v][k] = ok ? wk : 0.f;
            // inside the staged box?  (always, for a valid tap, unless the slack was exceeded)
            const bool inbox = ok && cx >= b0[0] && cx < b0[0] + bxs && cy >= b0[1] && cy < b0[1] + bys &&
                               cz >= b0[2] && cz < b0[2] + bzs;
            miss = miss || (ok && !inbox);
            const unsigned int lo16 = inbox ? (unsigned int)(base + (k >> 2) * bys * FBBEV_HW_PITCH + ((k >> 1) & 1) * FBBEV_HW_PITCH + (k & 1)) : 0u;
            loff[v][k >> 1] |= lo16 << (16 * (k & 1));
        }
    }
    // one valid tap outside its box anywhere in the workgroup -> the whole brick gathers from global memory
    __shared__ int any_miss;
    if (tid == 0) any_miss = 0;
    __syncthreads();
    if (miss) any_miss = 1;
    __syncthreads();
    use_lds = use_lds && any_miss == 0;
    const int c0 = grp * ch_per_block;
    const int c1 = (c0 + ch_per_block < CH) ? c0 + ch_per_block : CH;
    const int rows = bzs * bys;
    for (int c = c0; c < c1; ++c) {
        const long long so = (long long)b * hist_stride_b + (long long)c * ZYX;
        const long long dof = (long long)b * out_stride_b + (long long)c * ZYX;
        if (use_lds && !empty) {
            __syncthreads();                                       // the previous channel's taps are done
            for (int r = wave; r < rows; r += 16) {                // a wave stages whole rows: coalesced along x
                const int bz = r / bys, by = r - bz * bys;
                const long long g = so + ((long long)(b0[2] + bz) * Y + (b0[1] + by)) * X + b0[0];
                for (int xx = lane; xx < bxs; xx += 64) box[r * FBBEV_HW_PITCH + xx] = fbbev_ld_elem<ET>(hist, g + xx);
            }
            __syncthreads();
        }
        if (use_lds) {
#pragma unroll
            for (int v = 0; v < FBBEV_HW_VPT; ++v) {
                fbbev_sched_fence();                      // one voxel's 8 LDS taps at a time: 16 waves per CU hide the latency
                if (vox[v] < 0) continue;
                float s0 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float a = empty ? 0.f : box[(loff[v][k >> 1] >> (16 * (k & 1))) & 0xffffu];
                    s0 = fmaf(a, w[v][k], s0);
                }
                fbbev_st_elem<ET>(out, dof + vox[v], s0);
            }
        } else {
#pragma unroll
            for (int v = 0; v < FBBEV_HW_VPT; ++v) {     // rare path (large rotations): the tap positions are re-derived here
                fbbev_sched_fence();                      // so that the common path carries no state for it
                if (vox[v] < 0) continue;
                const int x = vox[v] % X, y = (vox[v] / X) % Y, z = vox[v] / YX;
                float ix, iy, iz;
                src((float)x, (float)y, (float)z, ix, iy, iz);
                const bool fin = (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f) && (fabsf(iz) < 1.0e9f);
                const int x0 = fin ? (int)floorf(ix) : -2, y0 = fin ? (int)floorf(iy) : -2, z0 = fin ? (int)floorf(iz) : -2;
                float s0 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int cx = x0 + (k & 1), cy = y0 + ((k >> 1) & 1), cz = z0 + (k >> 2);
                    const bool ok = cx >= 0 && cx < X && cy >= 0 && cy < Y && cz >= 0 && cz < Z;
                    s0 = fmaf(fbbev_ld_elem<ET>(hist, so + (ok ? (cz * Y + cy) * X + cx : 0)), w[v][k], s0);
                }
                fbbev_st_elem<ET>(out, dof + vox[v], s0);
            }
        }
    }
}
