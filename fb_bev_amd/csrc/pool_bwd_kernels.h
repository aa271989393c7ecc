// pool_bwd_kernels.h -- training backward of the fused lift-splat, with no host sync and no re-sort.
//
// Replaces QuickCumsumCuda.backward (mmdet3d/ops/bev_pool_v2/bev_pool.py:39-78): the reference
// argsorts ranks_feat on the host side of autograd, rebuilds the intervals with boolean masks (two
// syncs), permutes the (B,C,Z,Y,X) gradient to channels-last with a full copy and only then launches
// bev_pool_v2_grad_kernel (src/bev_pool_cuda.cu:52-100).  Here the frustum's own structure is the
// index: the points of feature pixel (b,n,h,w) are exactly its D depth bins, point id
// ((b*N+n)*D+d)*H*W+hw, so no sort by ranks_feat is needed.
//
//   k_point_row_table : voxel-sorted position p -> its interval j (upper bound over interval_starts);
//                       table[ranks_depth[p]] = j, dropped points keep -1.
//   k_pool_bwd_rows   : per tile of TV voxels of a (b,z) plane that holds at least one interval, read
//                       the (C x TV) block of the NCZYX gradient coalesced along x, transpose it through
//                       LDS and write one C-float row per interval: rows[j][:] = out_grad[b,:,z,y,x].
//                       Empty tiles are never read.
//   k_pool_bwd_pixel  : half a wave64 per feature pixel, lanes over channels (CPL each).  The D table
//                       entries of the pixel are loaded one per lane; kept bins are visited in ascending
//                       d: depth_grad[pid] = <rows[j], feat[pixel]> (half-wave reduction), feat_grad[pixel]
//                       = in-order fmaf chain of rows[j] * depth[pid] -- the arithmetic of
//                       bev_pool_cuda.cu:77-98 with the points of a pixel taken in ascending depth-bin
//                       order (the reference's order is whatever its unstable argsort produced).
//                       Every element of depth_grad and feat_grad is written exactly once, zeros
//                       included: no zeros_like passes.
#pragma once
#include "rt.h"

__global__ void __launch_bounds__(256)
k_point_row_table(const int* __restrict__ ranks_depth, const int* __restrict__ interval_starts,
                  const int* __restrict__ counts, int n_intervals_max, int* __restrict__ table) {
    const int P = counts[0];
    int I = counts[1];
    if (I > n_intervals_max) I = n_intervals_max;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = I;                       // first interval whose start is > p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (interval_starts[mid] <= (int)p) lo = mid + 1; else hi = mid;
        }
        table[ranks_depth[p]] = lo - 1;
    }
}

template <int TV>
__global__ void __launch_bounds__(256)
k_pool_bwd_rows(int C, int Z, int YX, int tiles_per_plane, long long stride_b, long long stride_c,
                const float* __restrict__ out_grad, const int* __restrict__ interval_rank,
                const int* __restrict__ tile_meta, float* __restrict__ rows, const float* __restrict__ zgrad, float zscale) {
    // zgrad (B, C, Y*X), may be null: a gradient every z plane receives on top of out_grad, scaled -- the backward of the
    // training path's Z-mean (fbocc.py:359: d mean / d voxel = 1 / Z) folded into this read instead of an expand + add over
    // the whole volume
    constexpr int LD = TV + 4;
    constexpr int Q4 = TV / 4;
    float* tile = fbbev_dyn_lds_f32();            // [C][LD]
    const int t = blockIdx.x, tid = threadIdx.x;
    const int i0 = tile_meta[2 * t], i1 = tile_meta[2 * t + 2];
    if (i0 == i1) return;                         // block-uniform
    const int plane = t / tiles_per_plane, k = t - plane * tiles_per_plane;
    const int b = plane / Z, z = plane - b * Z;
    const int v0 = k * TV;
    const int nv = (YX - v0 < TV) ? (YX - v0) : TV;
    const float* __restrict__ base = out_grad + (long long)b * stride_b + (long long)z * YX + v0;
    const int n4 = C * Q4;
    // round 6: the tile's pieces are REQUESTED in batches of U before the first is used (written as one load-add-store per iteration
    // the loop compiled to load, s_waitcnt vmcnt(0), ds_write: ten round trips in a row per workgroup at C = 80 -- the pattern
    // tools/isa_waits.py found in the row kernels in round 5; 0.25 ms for an 0.8 GB read at BASELINE configs[2], B = 4)
    constexpr int U = 5;
    for (int idx0 = tid; idx0 < n4; idx0 += 256 * U) {
        fbbev_v4f v[U], zg[U];
        int cc[U], jj[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx_ = idx0 + 256 * u, idx = idx_ < n4 ? idx_ : idx0;
            cc[u] = idx / Q4; jj[u] = (idx - cc[u] * Q4) * 4;
            ok[u] = idx_ < n4 && jj[u] < nv;
            const int jl = jj[u] < nv ? jj[u] : 0;                                  // (clamped: unconditional loads)
            v[u] = *reinterpret_cast<const fbbev_v4f*>(base + (long long)cc[u] * stride_c + jl);
            if (zgrad) zg[u] = *reinterpret_cast<const fbbev_v4f*>(zgrad + ((long long)b * C + cc[u]) * YX + v0 + jl);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            fbbev_v4f w = v[u];
            if (zgrad) w += zg[u] * zscale;
            *reinterpret_cast<fbbev_v4f*>(tile + cc[u] * LD + jj[u]) = w;
        }
    }
    __syncthreads();
    const int rank0 = plane * YX + v0;
    const int C4 = C >> 2;
    const int work = (i1 - i0) * C4;
    constexpr int U2 = 4;
    for (int idx0 = tid; idx0 < work; idx0 += 256 * U2) {
        int tvs[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {                     // the intervals' voxel ranks first: one round trip per batch
            const int idx_ = idx0 + 256 * u, idx = idx_ < work ? idx_ : idx0;
            tvs[u] = interval_rank[i0 + idx / C4] - rank0;
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int idx = idx0 + 256 * u;
            if (idx >= work) continue;
            const int ii = idx / C4, c4 = idx - ii * C4;
            const float* col = tile + (4 * c4) * LD + tvs[u];
            fbbev_v4f r;
            r[0] = col[0]; r[1] = col[LD]; r[2] = col[2 * LD]; r[3] = col[3 * LD];
            *reinterpret_cast<fbbev_v4f*>(rows + (long long)(i0 + ii) * C + 4 * c4) = r;
        }
    }
}

// CPL channels per lane (4: C <= 128, 8: C <= 256); U kept bins in flight per half-wave.
template <int CPL>
__global__ void __launch_bounds__(256)
k_pool_bwd_pixel(int C, int D, int HW, long long n_pixels, long long row_stride,
                 const float* __restrict__ rows, const float* __restrict__ depth,
                 const float* __restrict__ feat, const int* __restrict__ table,
                 float* __restrict__ depth_grad, float* __restrict__ feat_grad) {
    constexpr int U = 4;
    constexpr int V4 = CPL / 4;
    const int lane = threadIdx.x & 63, h = lane >> 5, l = lane & 31;
    const long long f = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool live = f < n_pixels;
    const long long bn = live ? f / HW : 0;
    const int hw = live ? (int)(f - bn * HW) : 0;
    const bool chan = live && l * CPL < C;
    fbbev_v4f fr[V4], g[V4];
#pragma unroll
    for (int q = 0; q < V4; ++q) {
        fr[q] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        g[q] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
        if (chan) fr[q] = *reinterpret_cast<const fbbev_v4f*>(feat + f * C + l * CPL + 4 * q);
    }
    for (int d0 = 0; d0 < D; d0 += 32) {
        const int d = d0 + l;
        const bool valid = live && d < D;
        const long long pid = valid ? (bn * D + d) * HW + hw : 0;
        const int row = valid ? table[pid] : -1;
        const float dep = (row >= 0) ? depth[pid] : 0.f;
        float mygrad = 0.f;
        const unsigned long long bal = __ballot(row >= 0);
        unsigned int m = (unsigned int)(bal >> (32 * h));
        const int n0 = __popcll(bal & 0xffffffffull), n1 = __popcll(bal >> 32);
        const int trips = ((n0 > n1 ? n0 : n1) + U - 1) / U;     // wave-uniform
        for (int it = 0; it < trips; ++it) {
            bool has[U];
            int src[U], r[U];
            float dp[U];
            fbbev_v4f og[U][V4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                has[u] = m != 0u;
                src[u] = has[u] ? (__ffsll((long long)m) - 1) : 0;
                m &= m - 1u;                                      // 0 stays 0
                r[u] = __shfl(row, 32 * h + src[u], 64);
                dp[u] = __shfl(dep, 32 * h + src[u], 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < V4; ++q) {
                    og[u][q] = fbbev_v4f{0.f, 0.f, 0.f, 0.f};
                    if (has[u] && chan)
                        og[u][q] = *reinterpret_cast<const fbbev_v4f*>(rows + (long long)r[u] * row_stride + l * CPL + 4 * q);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float p = 0.f;
                if (has[u] && chan) {
#pragma unroll
                    for (int q = 0; q < V4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            p = fmaf(og[u][q][e], fr[q][e], p);
                            g[q][e] = fmaf(og[u][q][e], dp[u], g[q][e]);
                        }
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);   // stays inside the half-wave
                if (has[u] && l == src[u]) mygrad = p;
            }
        }
        if (valid) depth_grad[pid] = mygrad;
    }
    if (chan) {
#pragma unroll
        for (int q = 0; q < V4; ++q)
            *reinterpret_cast<fbbev_v4f*>(feat_grad + f * C + l * CPL + 4 * q) = g[q];
    }
}
