// geom_kernels.h -- frustum lift geometry: image-plane frustum -> ego-frame points.
//
// Replaces LSSViewTransformerFunction3D.get_lidar_coor
// (fbbev/view_transformation/forward_projection/view_transformer.py:458-498): in the reference two
// batched 3x3 torch.inverse calls, three broadcast matmuls and a cat, ~8 launches that materialise
// three (B,N,D,H,W,3) temporaries.  Here one kernel: the per-camera 3x3 algebra is done once per
// workgroup in LDS, then every lane transforms points with the same operation order as the
// reference (subtract post_trans, inv(post_rots)*p, scale xy by depth, (rots*inv(K))*p + trans,
// bda*p).  The 3x3 inverses are closed-form (adjugate / determinant) instead of LU, so `coor` can
// differ from the torch result in the last ulps: the bit-exactness contract of the path is pinned
// at voxel_pooling_prepare_v2's INPUT.
#pragma once
#include "rt.h"

__device__ __forceinline__ void fbbev_inv3(const float* m, float* o) {
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    const float r = 1.0f / det;
    o[0] = A * r; o[1] = -(b * i - c * h) * r; o[2] = (b * f - c * e) * r;
    o[3] = B * r; o[4] = (a * i - c * g) * r;  o[5] = -(a * f - c * d) * r;
    o[6] = C * r; o[7] = -(a * h - b * g) * r; o[8] = (a * e - b * d) * r;
}

__device__ __forceinline__ void fbbev_mat3(const float* a, const float* b, float* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

struct fbbev_cam_ptrs {
    const float *xs, *ys, *ds;                               // frustum axes (W), (H), (D)
    const float *rots, *trans, *intrins, *post_rots, *post_trans, *bda;
    int N, D, H, W;
};

// per-camera algebra, done once per workgroup by one thread into m[33]:
//   m[0:9] inv(post_rots), m[9:18] rots*inv(K), m[18:27] bda, m[27:30] post_trans, m[30:33] trans
__device__ __forceinline__ void fbbev_cam_setup(const fbbev_cam_ptrs& g, int cam, float* m) {
    const int b = cam / g.N;
    float ik[9];
    fbbev_inv3(g.post_rots + cam * 9, m);
    fbbev_inv3(g.intrins + cam * 9, ik);
    fbbev_mat3(g.rots + cam * 9, ik, m + 9);
    for (int j = 0; j < 9; ++j) m[18 + j] = g.bda[b * 9 + j];
    for (int j = 0; j < 3; ++j) { m[27 + j] = g.post_trans[cam * 3 + j]; m[30 + j] = g.trans[cam * 3 + j]; }
}

// one frustum point (u, v, depth) -> ego frame; the SAME expression sequence serves the materialising
// kernel (k_lidar_coor) and the fused geometry->key source of the sort's first pass, so both produce the
// same bits.
__device__ __forceinline__ void fbbev_point_coor(const float* m, float u, float v, float d, float& ox, float& oy,
                                                 float& oz) {
    float px = u - m[27], py = v - m[28], pz = d - m[29];
    float qx = m[0] * px + m[1] * py + m[2] * pz;
    float qy = m[3] * px + m[4] * py + m[5] * pz;
    float qz = m[6] * px + m[7] * py + m[8] * pz;
    qx *= qz; qy *= qz;
    px = m[9] * qx + m[10] * qy + m[11] * qz + m[30];
    py = m[12] * qx + m[13] * qy + m[14] * qz + m[31];
    pz = m[15] * qx + m[16] * qy + m[17] * qz + m[32];
    ox = m[18] * px + m[19] * py + m[20] * pz;
    oy = m[21] * px + m[22] * py + m[23] * pz;
    oz = m[24] * px + m[25] * py + m[26] * pz;
}

// grid: (ceil(D*H*W / 256), B*N) flattened into blockIdx.x = cam * chunks + chunk
__global__ void __launch_bounds__(256)
k_lidar_coor(fbbev_cam_ptrs g, int chunks, float* __restrict__ coor) {
    __shared__ float m[33];
    const int cam = blockIdx.x / chunks, chunk = blockIdx.x - cam * chunks;
    if (threadIdx.x == 0) fbbev_cam_setup(g, cam, m);
    __syncthreads();
    const int dhw = g.D * g.H * g.W;
    const int i = chunk * 256 + threadIdx.x;
    if (i < dhw) {
        const int w = i % g.W, h = (i / g.W) % g.H, d = i / (g.W * g.H);
        float ox, oy, oz;
        fbbev_point_coor(m, g.xs[w], g.ys[h], g.ds[d], ox, oy, oz);
        float* o = coor + ((long long)cam * dhw + i) * 3;
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
}

// All levels of the camera-token pyramid in ONE launch (round 4): bevformer.py:95-117 runs the flatten / permute / + cams_embeds per
// level and concatenates; one k_nchw_to_nhwc launch per level cost ~13 us each for a few hundred tokens (latency of a nearly empty
// launch), four per forward.  Blocks [blk0[l], blk0[l+1]) transpose level l (n_images x C x hw[l]) into rows out_off[l] .. of every
// image's token block; bias row (img % bias_rows) is added (cams_embeds).
struct fbbev_token_levels {
    const float* in[8];
    long long out_off[8];       // float offset of the level's first token inside an image's block (level_start * C)
    int hw[8];
    int blk0[9];
    int n;
};
__global__ void __launch_bounds__(256)
k_nchw_to_nhwc_levels(fbbev_token_levels lv, float* __restrict__ out, int C, int tiles_c, long long out_image_stride,
                      const float* __restrict__ bias, int bias_rows) {
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < lv.n && (int)blockIdx.x >= lv.blk0[l + 1]) ++l;
    const int HW = lv.hw[l], tiles_hw = (HW + 31) / 32;
    const int t = (int)blockIdx.x - lv.blk0[l];
    const int img = t / (tiles_c * tiles_hw), r = t - img * (tiles_c * tiles_hw);
    const int tc = r / tiles_hw, th = r - tc * tiles_hw;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float* src = lv.in[l] + (long long)img * C * HW;
    float* dst = out + (long long)img * out_image_stride + lv.out_off[l];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tc * 32 + ly + 8 * k, p = th * 32 + lx;
        tile[ly + 8 * k][lx] = (c < C && p < HW) ? src[(long long)c * HW + p] : 0.f;
    }
    __syncthreads();
    const int cb = tc * 32 + lx;
    const float add = (bias != nullptr && cb < C) ? bias[(long long)(img % bias_rows) * C + cb] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = th * 32 + ly + 8 * k;
        if (cb < C && p < HW) fbbev_st(dst + (long long)p * C + cb, tile[lx][ly + 8 * k] + add);
    }
}

// ---------------------------------------------------------------- BEV voxel centres -> image points
// Replaces bevformer_encoder.point_sampling
// (fbbev/view_transformation/backward_projection/bevformer_utils/bevformer_encoder.py:91-120): three
// batched 3x3 inverses + three broadcast matmuls over B*N*Y*X*Za points (hipBLASLt runs those 3x1
// "GEMMs" at ~3 ms each on the shipped config) + ~20 elementwise launches.  One kernel: per-camera
// algebra once per workgroup, then per point
//   p = inv(bda)*p - trans ; c = inv(rots*inv(K))*p ; (c.xy /= max(c.z, eps)) ; c = post_rots*c + post_trans
//   u = c.x/W_in, v = c.y/H_in ; mask = c.z > eps and eps < u,v < 1-eps
// Outputs in the layouts the encoder uses: ref_cam (N,B,Q,Za,2), mask (N,B,Q,Za) u8, qdepth (N,B,Q,Za).
__global__ void __launch_bounds__(256)
k_point_sampling(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                 const float* __restrict__ rots, const float* __restrict__ trans,
                 const float* __restrict__ intrins, const float* __restrict__ post_rots,
                 const float* __restrict__ post_trans, const float* __restrict__ bda, int B, int N, int Y, int X,
                 int Za, float ogfH, float ogfW, int chunks, float* __restrict__ ref_cam,
                 unsigned char* __restrict__ mask, float* __restrict__ qdepth, int ppt) {
    __shared__ float m[9 + 9 + 9 + 3 + 3];  // inv(bda), inv(rots*inv(K)), post_rots, trans, post_trans
    const int cam_b = blockIdx.x / chunks, chunk = blockIdx.x - cam_b * chunks;   // cam_b = b*N + n
    const int b = cam_b / N, n = cam_b - b * N;
    if (threadIdx.x == 0) {
        float ik[9], comb[9];
        fbbev_inv3(bda + b * 9, m);
        fbbev_inv3(intrins + cam_b * 9, ik);
        fbbev_mat3(rots + cam_b * 9, ik, comb);
        fbbev_inv3(comb, m + 9);
        for (int j = 0; j < 9; ++j) m[18 + j] = post_rots[cam_b * 9 + j];
        for (int j = 0; j < 3; ++j) { m[27 + j] = trans[cam_b * 3 + j]; m[30 + j] = post_trans[cam_b * 3 + j]; }
    }
    __syncthreads();
    const float eps = 1e-5f;
    const int npts = Y * X * Za;
    // round 5: `ppt` points per thread (the launcher's choice: as many as keep >= 1024 workgroups).  The per-camera algebra above is one
    // thread's serial work behind its own loads -- with one point per thread it was most of a workgroup's life (33 us for 45 MB at configs[2])
    for (int k = 0; k < ppt; ++k) {
        const int i = (chunk * ppt + k) * 256 + (int)threadIdx.x;
        if (i >= npts) break;
        const int z = i % Za, x = (i / Za) % X, y = i / (Za * X);
        const float px = xs[x], py = ys[y], pz = zs[z];
        float qx = m[0] * px + m[1] * py + m[2] * pz - m[27];
        float qy = m[3] * px + m[4] * py + m[5] * pz - m[28];
        float qz = m[6] * px + m[7] * py + m[8] * pz - m[29];
        float cx = m[9] * qx + m[10] * qy + m[11] * qz;
        float cy = m[12] * qx + m[13] * qy + m[14] * qz;
        const float cz = m[15] * qx + m[16] * qy + m[17] * qz;
        const float den = fmaxf(cz, eps);
        cx = cx / den; cy = cy / den;
        const float ux = m[18] * cx + m[19] * cy + m[20] * cz + m[30];
        const float uy = m[21] * cx + m[22] * cy + m[23] * cz + m[31];
        const float uz = m[24] * cx + m[25] * cy + m[26] * cz + m[32];
        const float u = ux / ogfW, v = uy / ogfH;
        const bool ok = (uz > eps) && (u > eps) && (u < (1.0f - eps)) && (v > eps) && (v < (1.0f - eps));
        const long long o = ((long long)(n * B + b)) * npts + i;   // (N,B,Q=Y*X,Za)
        fbbev_st(reinterpret_cast<fbbev_v2f*>(ref_cam + o * 2), fbbev_v2f{u, v});
        fbbev_st(mask + o, (unsigned char)(ok ? 1 : 0));
        fbbev_st(qdepth + o, uz);
    }
}

// ---------------------------------------------------------------- context (B*N,C,H*W) -> feat (B*N,H*W,C)
// The depth net emits NCHW; bev_pool_v2 gathers channel rows, so the reference makes feat.permute(0,1,3,4,2)
// contiguous inside the op (bev_pool.py:18, view_transformer.py:536).  LDS-tiled 32x32 transpose: both the
// read (along H*W) and the write (along C) are coalesced.
// Generalised for the backward projection's camera tokens (bevformer.py:95-117: flatten(3).permute(1,0,3,2) + cams_embeds,
// cat over levels, and the per-camera rebatch permute of spatial_cross_attention_depth.py:151): image `img` is written at
// out + img*out_image_stride + out_offset, with bias[(img % bias_rows)*C + c] added when bias != nullptr.
__global__ void __launch_bounds__(256)
k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int tiles_c, int tiles_hw,
               long long out_image_stride, long long out_offset, const float* __restrict__ bias, int bias_rows,
               const float* __restrict__ pos_bias) {
    __shared__ float tile[32][33];
    const int t = blockIdx.x;
    const int img = t / (tiles_c * tiles_hw), r = t - img * (tiles_c * tiles_hw);
    const int tc = r / tiles_hw, th = r - tc * tiles_hw;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;           // 32 x 8
    const float* src = in + (long long)img * C * HW;
    float* dst = out + (long long)img * out_image_stride + out_offset;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tc * 32 + ly + 8 * k, p = th * 32 + lx;
        tile[ly + 8 * k][lx] = (c < C && p < HW) ? src[(long long)c * HW + p] : 0.f;
    }
    __syncthreads();
    const int cb = tc * 32 + lx;
    if (pos_bias != nullptr) {                      // + a per-position row (HW, C), the same for every image: the learned BEV embedding
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = th * 32 + ly + 8 * k;
            if (cb < C && p < HW) fbbev_st(dst + (long long)p * C + cb, tile[lx][ly + 8 * k] + pos_bias[(long long)p * C + cb]);
        }
        return;
    }
    if (bias == nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = th * 32 + ly + 8 * k;
            if (cb < C && p < HW) fbbev_st(dst + (long long)p * C + cb, tile[lx][ly + 8 * k]);
        }
        return;
    }
    const float add = cb < C ? bias[(long long)(img % bias_rows) * C + cb] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = th * 32 + ly + 8 * k;
        if (cb < C && p < HW) fbbev_st(dst + (long long)p * C + cb, tile[lx][ly + 8 * k] + add);
    }
}
