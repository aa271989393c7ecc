// capi.hip -- extern "C" entry points of libfbbev_hip.so (include/fbbev.h): argument checks,
// launch geometry, workspace carving.  No torch, no host synchronisation, caller's stream.
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "rt.h"
#include "pool_kernels.h"
#include "pool_bwd_kernels.h"
#include "rank_kernels.h"
#include "sort_kernels.h"
#include "msda_kernels.h"
#include "geom_kernels.h"
#include "da_kernels.h"
#include "history_kernels.h"
#include "norm_kernels.h"
#include "history_conv_kernels.h"
#include "history_fused_kernels.h"
#include "history_conv_x3_kernels.h"
#include "history_fused_x3_kernels.h"
#include "rows_linear_kernels.h"
#include "da_fused_kernels.h"
#include "da_bwd_planes_kernels.h"
#include "msda_bwd_kernels.h"
#include "conv3d_kernels.h"
#include "../../include/fbbev.h"

#include "capi_common.h"

extern "C" int fbbev_version(void) { return 100; }

// ------------------------------------------------------------------------------ bev_pool_v2 fwd
extern "C" int fbbev_bev_pool_v2_fwd(int c, int n_intervals, const float* depth, const float* feat,
                                     const int32_t* ranks_depth, const int32_t* ranks_feat,
                                     const int32_t* ranks_bev, const int32_t* interval_starts,
                                     const int32_t* interval_lengths, float* out,
                                     fbbev_stream_t stream_) {
    if (c <= 0 || n_intervals < 0) return FBBEV_E_BADARG;
    if (n_intervals == 0) return 0;
    if (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_starts ||
        !interval_lengths || !out) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    if (c % 4 == 0 && aligned16(feat) && aligned16(out)) {
        const long long threads = (long long)n_intervals * (c / 4);
        FBBEV_LAUNCH(k_pool_fwd_rows<4>, (threads + 255) / 256, 256, 0, stream, c, n_intervals, depth,
                     feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out);
    } else {
        const long long threads = (long long)n_intervals * c;
        FBBEV_LAUNCH(k_pool_fwd_rows<1>, (threads + 255) / 256, 256, 0, stream, c, n_intervals, depth,
                     feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out);
    }
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ bev_pool_v2 bwd
extern "C" int fbbev_bev_pool_v2_bwd(int c, int n_intervals, const float* out_grad,
                                     const float* depth, const float* feat,
                                     const int32_t* ranks_depth, const int32_t* ranks_feat,
                                     const int32_t* ranks_bev, const int32_t* interval_starts,
                                     const int32_t* interval_lengths, float* depth_grad,
                                     float* feat_grad, fbbev_stream_t stream_) {
    if (c <= 0 || n_intervals < 0) return FBBEV_E_BADARG;
    if (c > 256) return FBBEV_E_UNSUPPORTED;
    if (n_intervals == 0) return 0;
    if (!out_grad || !depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
        !interval_starts || !interval_lengths || !depth_grad || !feat_grad) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    const long long blocks = ((long long)n_intervals * 64 + 255) / 256;  // one wave64 per interval
    const int nch = (c + 63) / 64;
#define FBBEV_BWD(NCH)                                                                             \
    FBBEV_LAUNCH(k_pool_bwd<NCH>, blocks, 256, 0, stream, c, n_intervals, out_grad, depth, feat,   \
                 ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, depth_grad, \
                 feat_grad)
    switch (nch) {
        case 1: FBBEV_BWD(1); break;
        case 2: FBBEV_BWD(2); break;
        case 3: FBBEV_BWD(3); break;
        default: FBBEV_BWD(4); break;
    }
#undef FBBEV_BWD
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ frustum geometry
extern "C" int fbbev_lidar_coor(const float* xs, const float* ys, const float* ds, const float* rots,
                                const float* trans, const float* intrins, const float* post_rots,
                                const float* post_trans, const float* bda, int B, int N, int D, int H,
                                int W, float* coor, fbbev_stream_t stream_) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return FBBEV_E_BADARG;
    if (!xs || !ys || !ds || !rots || !trans || !intrins || !post_rots || !post_trans || !bda || !coor)
        return FBBEV_E_BADARG;
    const long long dhw = (long long)D * H * W;
    const long long chunks = (dhw + 255) / 256;
    if (chunks * B * N >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    fbbev_cam_ptrs g;
    g.xs = xs; g.ys = ys; g.ds = ds; g.rots = rots; g.trans = trans; g.intrins = intrins; g.post_rots = post_rots;
    g.post_trans = post_trans; g.bda = bda; g.N = N; g.D = D; g.H = H; g.W = W;
    FBBEV_LAUNCH(k_lidar_coor, chunks * B * N, 256, 0, (fbbev_rt_stream)stream_, g, (int)chunks, coor);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_nchw_to_nhwc(const float* in, float* out, int n_images, int C, int HW, fbbev_stream_t stream_) {
    return fbbev_tokens_from_nchw(in, out, n_images, C, HW, (long long)C * HW, 0, nullptr, 0, stream_);
}

static int tokens_from_nchw_impl(const float* in, float* out, int n_images, int C, int HW, long long out_image_stride,
                                 long long out_offset, const float* bias, int bias_rows, const float* pos_bias,
                                 fbbev_stream_t stream_) {
    if (n_images < 0 || C <= 0 || HW <= 0 || out_offset < 0 || out_image_stride < (long long)C * HW) return FBBEV_E_BADARG;
    if (bias && bias_rows <= 0) return FBBEV_E_BADARG;
    if (n_images == 0) return 0;
    if (!in || !out) return FBBEV_E_BADARG;
    const int tc = (C + 31) / 32, th = (HW + 31) / 32;
    const long long blocks = (long long)n_images * tc * th;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_nchw_to_nhwc, blocks, 256, 0, (fbbev_rt_stream)stream_, in, out, C, HW, tc, th, out_image_stride,
                 out_offset, bias, bias ? bias_rows : 1, pos_bias);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_tokens_from_nchw(const float* in, float* out, int n_images, int C, int HW,
                                      long long out_image_stride, long long out_offset, const float* bias,
                                      int bias_rows, fbbev_stream_t stream_) {
    return tokens_from_nchw_impl(in, out, n_images, C, HW, out_image_stride, out_offset, bias, bias_rows, nullptr, stream_);
}

// every level of the camera-token pyramid in one launch: level l = in[l] (n_images, C, hw[l]) -> rows [start_l, start_l + hw[l]) of
// every image's (sum hw, C) token block in `out`, start_l = hw[0] + .. + hw[l-1]; + bias[(img % bias_rows), c] when bias is given
extern "C" int fbbev_tokens_from_nchw_levels(const float* const* in, const int32_t* hw, int n_levels, float* out, int n_images, int C,
                                             const float* bias, int bias_rows, fbbev_stream_t stream_) {
    if (n_levels <= 0 || n_levels > 8 || n_images < 0 || C <= 0 || !in || !hw) return FBBEV_E_BADARG;
    if (bias && bias_rows <= 0) return FBBEV_E_BADARG;
    if (n_images == 0) return 0;
    if (!out) return FBBEV_E_BADARG;
    fbbev_token_levels lv;
    const int tc = (C + 31) / 32;
    long long start = 0, blocks = 0;
    for (int l = 0; l < 8; ++l) { lv.in[l] = nullptr; lv.out_off[l] = 0; lv.hw[l] = 1; lv.blk0[l] = 0; }
    for (int l = 0; l < n_levels; ++l) {
        if (hw[l] <= 0 || !in[l]) return FBBEV_E_BADARG;
        lv.in[l] = in[l]; lv.hw[l] = hw[l]; lv.out_off[l] = start * C; lv.blk0[l] = (int)blocks;
        start += hw[l];
        blocks += (long long)n_images * tc * ((hw[l] + 31) / 32);
        if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    }
    lv.blk0[n_levels] = (int)blocks;
    lv.n = n_levels;
    FBBEV_LAUNCH(k_nchw_to_nhwc_levels, blocks, 256, 0, (fbbev_rt_stream)stream_, lv, out, C, tc, start * C, bias, bias ? bias_rows : 1);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// out[img, p, c] = in[img, c, p] + pos_bias[p, c]: the BEV queries of the backward projection (backward_projection.py:96-99:
// lss_bev flattened to tokens + the learned bev_embedding) in one transposing pass
extern "C" int fbbev_tokens_from_nchw_pos(const float* in, float* out, int n_images, int C, int HW, long long out_image_stride,
                                          long long out_offset, const float* pos_bias, fbbev_stream_t stream_) {
    if (!pos_bias) return FBBEV_E_BADARG;
    return tokens_from_nchw_impl(in, out, n_images, C, HW, out_image_stride, out_offset, nullptr, 0, pos_bias, stream_);
}

extern "C" int fbbev_point_sampling(const float* xs, const float* ys, const float* zs, const float* rots,
                                    const float* trans, const float* intrins, const float* post_rots,
                                    const float* post_trans, const float* bda, int B, int N, int Y, int X,
                                    int Za, float ogfH, float ogfW, float* ref_cam, uint8_t* mask,
                                    float* qdepth, fbbev_stream_t stream_) {
    if (B <= 0 || N <= 0 || Y <= 0 || X <= 0 || Za <= 0 || !(ogfH > 0.f) || !(ogfW > 0.f)) return FBBEV_E_BADARG;
    if (!xs || !ys || !zs || !rots || !trans || !intrins || !post_rots || !post_trans || !bda || !ref_cam ||
        !mask || !qdepth) return FBBEV_E_BADARG;
    const long long npts = (long long)Y * X * Za;
    int ppt = 8;                                            // points per thread: the most that leaves >= 1024 workgroups
    while (ppt > 1 && (npts + 256 * ppt - 1) / (256 * ppt) * B * N < 1024) ppt >>= 1;
    const long long chunks = (npts + 256 * ppt - 1) / (256 * ppt);
    if (chunks * B * N >= (1ll << 31) || ((uintptr_t)ref_cam & 7) != 0) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_point_sampling, chunks * B * N, 256, 0, (fbbev_rt_stream)stream_, xs, ys, zs, rots, trans,
                 intrins, post_rots, post_trans, bda, B, N, Y, X, Za, ogfH, ogfW, (int)chunks, ref_cam, mask,
                 qdepth, ppt);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ voxel ranking
static inline int key_bits(long long total_voxels) {
    int bits = 1;
    while ((1ll << bits) < total_voxels) ++bits;   // every rank < total_voxels <= 2^bits
    return bits;
}

// Chunk shapes of the ranking chain.  A workgroup owns one chunk; fat chunks keep the count matrix (chunks x digits)
// small enough to be summed directly by every scatter workgroup, and the chunk is the smallest one whose workgroup count
// still fits ONE round on the 256 CUs (a 1024-thread workgroup is alone on its CU: a second, partly filled round doubles
// the kernel's time -- measured: 325 workgroups 56 us, the same work in 146 workgroups 29 us).
//   variant: 0 = 4 waves x 8 (2048), 1 = 16 x 4 (4096), 2 = 16 x 8 (8192), 3 = 16 x 12 (12288), 4 = 16 x 16 (16384)
//            5 = 8 waves x 8 (4096), 6 = 8 x 16 (8192): 512-thread workgroups, TWO per CU -- the phases of a scatter workgroup
//            (column sums -> key loads -> ballot ranking -> scatter) are serial and barrier-separated; a second workgroup on
//            the CU fills them (round 3; chosen only through FBBEV_RANK_SHAPE until measured better)
static const int kSortTile[7] = {2048, 4096, 8192, 12288, 16384, 4096, 8192};
static int sort_variant_for(long long items) {
    for (int v = 0; v < 5; ++v)
        if ((items + kSortTile[v] - 1) / kSortTile[v] <= 256) return v;
    // more than one round of the fattest chunks: 512-thread workgroups of 8192 pairs, two per CU, interleave their phases
    // (measured at the shipped grid, B = 16, 5.4 M points: 0.262 -> 0.239 ms per build, profiles/r03_time_rank_shapes.jsonl)
    return 6;
}
//   interval kernels: 0 = 4 waves x 4 (1024), 1 = 16 x 4 (4096), 2 = 16 x 8 (8192)
static const int kIvTile[3] = {1024, 4096, 8192};
static int iv_variant_for(long long items) {
    for (int v = 0; v < 3; ++v)
        if ((items + kIvTile[v] - 1) / kIvTile[v] <= 256) return v;
    return 2;
}

// (m, sh) of fbbev_div for an invariant divisor d (Granlund & Montgomery, N = 32): l = ceil(log2 d),
// m = floor(2^32 (2^l - d) / d) + 1, sh = l - 1
static fbbev_fastdiv make_fastdiv(unsigned int d) {
    fbbev_fastdiv f;
    f.d = d; f.m = 0; f.sh = 0;
    if (d <= 1) { f.d = 1; return f; }
    unsigned int l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (unsigned int)((((1ull << l) - d) << 32) / d + 1);
    f.sh = l - 1;
    return f;
}

struct rank_ws_layout {
    size_t keys_a, keys_t, vals_t, matrix, ctot, chunk_info, total;
    int v0, wgs0;          // pass 0 (keys + scatter over all n points)
    int v1, wgs1;          // later passes (over the P kept pairs; P is only known on the device: sized for about n/2,
                           // the grid covers the worst case P = n and surplus workgroups leave at once)
    int vi, wgsi;          // interval kernels
};

static rank_ws_layout rank_layout(long long n) {
    rank_ws_layout L;
    L.v0 = sort_variant_for(n);
    L.v1 = sort_variant_for((n + 1) / 2);
    L.vi = iv_variant_for((n + 1) / 2);
    // tuning / test knob: FBBEV_RANK_SHAPE="v0,v1,vi" forces the chunk variants (results do not depend on them)
    if (const char* env = getenv("FBBEV_RANK_SHAPE")) {
        int a = -1, b = -1, c = -1;
        if (sscanf(env, "%d,%d,%d", &a, &b, &c) == 3 && a >= 0 && a < 7 && b >= 0 && b < 7 && c >= 0 && c < 3) {
            L.v0 = a; L.v1 = b; L.vi = c;
        }
    }
    L.wgs0 = (int)((n + kSortTile[L.v0] - 1) / kSortTile[L.v0]);
    L.wgs1 = (int)((n + kSortTile[L.v1] - 1) / kSortTile[L.v1]);
    L.wgsi = (int)((n + kIvTile[L.vi] - 1) / kIvTile[L.vi]);
    size_t off = 0;
    L.keys_a = off; off = align_up(off + (size_t)n * 4, 256);
    L.keys_t = off; off = align_up(off + (size_t)n * 4, 256);
    L.vals_t = off; off = align_up(off + (size_t)n * 4, 256);
    const int rows = L.wgs0 > L.wgs1 ? L.wgs0 : L.wgs1;
    // the segmented (per-sample) sort: at most n / TILE + 256 sample-aligned chunks (<= 256 samples), 2^10 digit columns, + ctot
    const size_t seg_rows = (size_t)(n / FBBEV_SEG_TILE) + 256 + 1;
    size_t mbytes = (size_t)rows * FBBEV_SORT_MAX_NB * 4;
    if (seg_rows * FBBEV_SEG_NB * 4 > mbytes) mbytes = seg_rows * FBBEV_SEG_NB * 4;
    L.matrix = off; off = align_up(off + mbytes, 256);
    L.ctot = off; off = align_up(off + seg_rows * 4, 256);
    L.chunk_info = off; off = align_up(off + (size_t)L.wgsi * 8, 256);
    L.total = off;                                         // no per-build state: nothing to clear, no kernel waits for another
    return L;
}

// one launch of KERNEL<threads or waves, per-thread items> for chunk variant v
#define FBBEV_SORT_DISPATCH(KERNEL, v, by_waves, grid, stream, ...)                                             \
    do {                                                                                                        \
        switch (v) {                                                                                            \
            case 0: FBBEV_LAUNCH((KERNEL<(by_waves) ? 4 : 256, 8>), grid, 256, 0, stream, __VA_ARGS__); break;   \
            case 1: FBBEV_LAUNCH((KERNEL<(by_waves) ? 16 : 1024, 4>), grid, 1024, 0, stream, __VA_ARGS__); break; \
            case 2: FBBEV_LAUNCH((KERNEL<(by_waves) ? 16 : 1024, 8>), grid, 1024, 0, stream, __VA_ARGS__); break; \
            case 3: FBBEV_LAUNCH((KERNEL<(by_waves) ? 16 : 1024, 12>), grid, 1024, 0, stream, __VA_ARGS__); break; \
            case 5: FBBEV_LAUNCH((KERNEL<(by_waves) ? 8 : 512, 8>), grid, 512, 0, stream, __VA_ARGS__); break;     \
            case 6: FBBEV_LAUNCH((KERNEL<(by_waves) ? 8 : 512, 16>), grid, 512, 0, stream, __VA_ARGS__); break;    \
            default: FBBEV_LAUNCH((KERNEL<(by_waves) ? 16 : 1024, 16>), grid, 1024, 0, stream, __VA_ARGS__); break; \
        }                                                                                                       \
    } while (0)

extern "C" size_t fbbev_rank_workspace_bytes(int64_t n_points) {
    if (n_points <= 0) return 256;
    return rank_layout(n_points).total;
}

static int rank_seg_read_env() { const char* e = getenv("FBBEV_RANK_SEG"); return e ? atoi(e) : -1; }
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: the tests switch modes inside one process
static int rank_seg_mode() { return rank_seg_read_env(); }
#else
static int rank_seg_mode() { static const int m = rank_seg_read_env(); return m; }      // read once per process
#endif

static int rank_build_impl(const float* coor, const fbbev_cam_ptrs* cams, const float* frustum, int B, int N, int D,
                           int H, int W,
                           const float* lower3, const float* interval3, const float* grid_size3,
                           int32_t* ranks_bev, int32_t* ranks_depth, int32_t* ranks_feat,
                           int32_t* interval_starts, int32_t* interval_lengths, int32_t* interval_rank,
                           int32_t* counts, void* workspace, size_t workspace_bytes, fbbev_rt_stream stream,
                           const float* point_depth = nullptr, float depth_thr = 0.f, const int* skip = nullptr) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return FBBEV_E_BADARG;
    if (point_depth && cams) return FBBEV_E_UNSUPPORTED;      // the depth filter belongs to the coor-based (two-step) contract
    if ((!coor && !cams) || !lower3 || !interval3 || !grid_size3 || !ranks_bev || !ranks_depth || !ranks_feat ||
        !interval_starts || !interval_lengths || !counts || !workspace) return FBBEV_E_BADARG;
    const long long n = (long long)B * N * D * H * W;
    if (n >= (1ll << 30)) return FBBEV_E_UNSUPPORTED;          // 30-bit counts in the look-back words
    const rank_ws_layout L = rank_layout(n);
    if (workspace_bytes < L.total) return FBBEV_E_WORKSPACE;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    unsigned int* keys_a = reinterpret_cast<unsigned int*>(ws + L.keys_a);
    unsigned int* keys_t = reinterpret_cast<unsigned int*>(ws + L.keys_t);
    unsigned int* vals_t = reinterpret_cast<unsigned int*>(ws + L.vals_t);
    int* matrix = reinterpret_cast<int*>(ws + L.matrix);
    int2* chunk_info = reinterpret_cast<int2*>(ws + L.chunk_info);

    fbbev_grid_params gp;
    gp.lx = lower3[0]; gp.ly = lower3[1]; gp.lz = lower3[2];
    gp.ix = interval3[0]; gp.iy = interval3[1]; gp.iz = interval3[2];
    gp.gx = grid_size3[0]; gp.gy = grid_size3[1]; gp.gz = grid_size3[2];
    // the 0-dim fp32 products of view_transformer.py:586-588, rounded step by step like torch
    volatile float zy = gp.gz * gp.gy;
    volatile float zyx = zy * gp.gx;
    volatile float yx = gp.gy * gp.gx;
    gp.f_zyx = zyx; gp.f_yx = yx;
    const long long total_voxels = (long long)B * (long long)gp.gz * (long long)gp.gy * (long long)gp.gx;
    if (total_voxels <= 0 || total_voxels >= (1ll << 30)) return FBBEV_E_UNSUPPORTED;
    // fp32 rank evaluation can round UP above 2^24 voxels: one spare bit covers total_voxels itself
    const int bits = key_bits(total_voxels + 128);
    struct { int passes, rb; } plan;
    plan.passes = (bits + FBBEV_SORT_MAX_RB - 1) / FBBEV_SORT_MAX_RB;
    plan.rb = (bits + plan.passes - 1) / plan.passes;          // digits as even as possible, <= 8 bits
    if (plan.rb < 4) plan.rb = 4;
    const int rb = plan.rb;
    // Per-sample (segmented) sort: B V <= 2^24 (the fp32 rank arithmetic is exact, so the keys of sample b lie in [b V, (b+1) V))
    // and V fits two 10-bit digits -- two passes instead of three, 6 launches instead of 9 (sort_kernels.h).  FBBEV_RANK_SEG=0
    // keeps the global sort (A/B timing, tests); the index tensors are the same either way.
    const int seg_mode = rank_seg_mode();                   // -1 auto | 0 never | 2 whenever legal (tests)
    const long long vps = total_voxels / B;
    const long long npb = n / B;
    const int lbits = key_bits(vps);
    const bool seg_legal = B <= 256 && total_voxels <= (1ll << 24) && lbits <= 2 * FBBEV_SEG_RB;
    // auto: when it saves a pass AND the launch has the shape the kernels are good at -- enough sample-aligned chunks to fill
    // the part (>= 96 workgroups) and few chunks per sample (every scatter workgroup sums the <= 64 count rows of its sample).
    // Measured (profiles/r03_time_rank_seg*.jsonl, whole build, eager): shipped grid B = 16  0.258 -> 0.190 ms, BL2 B = 16
    // 0.182 -> 0.175 ms, BL2 B = 4  0.133 -> 0.122 ms; it LOSES for one fat sample (BL1 B = 1: 487 chunks per sample,
    // 0.152 -> 0.258 ms) and for a single small sample (shipped grid B = 1: 42 workgroups), which keep the global sort.
    const long long seg_cps = (npb + FBBEV_SEG_TILE - 1) / FBBEV_SEG_TILE;
    const bool seg = seg_legal && seg_mode != 0 &&
                     (seg_mode == 2 || ((lbits + FBBEV_SEG_RB - 1) / FBBEV_SEG_RB < plan.passes && seg_cps <= 64 && B * seg_cps >= 96));
    if (seg) {
        fbbev_seg sg;
        sg.npb = (int)npb; sg.cps = (int)((npb + FBBEV_SEG_TILE - 1) / FBBEV_SEG_TILE); sg.vps = (unsigned int)vps; sg.nseg = B;
        const int passes = (lbits + FBBEV_SEG_RB - 1) / FBBEV_SEG_RB;                     // 1 or 2
        const int srb = (lbits + passes - 1) / passes < 4 ? 4 : (lbits + passes - 1) / passes;
        const int wgs = B * sg.cps;
        int* ctot = reinterpret_cast<int*>(ws + L.ctot);
        fbbev_geom_src gs;
        if (cams) { gs.cam = *cams; gs.frustum = frustum; gs.gp = gp; }
        else { gs.cam = fbbev_cam_ptrs(); gs.frustum = nullptr; gs.gp = gp; }
        if (cams) FBBEV_LAUNCH((k_keys_hist_seg<true>), wgs, FBBEV_SEG_NT, 0, stream, gs, (const float*)nullptr, gp, (const float*)nullptr, 0.f,
                               sg, srb, skip, keys_a, matrix, ctot);
        else FBBEV_LAUNCH((k_keys_hist_seg<false>), wgs, FBBEV_SEG_NT, 0, stream, gs, coor, gp, point_depth, depth_thr, sg, srb, skip, keys_a,
                          matrix, ctot);
        FBBEV_CHECK_LAUNCH();
        const unsigned int* kin = keys_a;
        const unsigned int* vin = nullptr;
        static const int pairs = [] { const char* e = getenv("FBBEV_RANK_PAIRS"); return e ? atoi(e) : 1; }();   // A/B knob, read once
        for (int p = 0; p < passes; ++p) {
            const bool to_out = p == passes - 1;
            // the intermediate (pass 0 -> pass 1) lives in keys_t | vals_t, which are adjacent: as ONE array of (key, value) pairs
            const int pm = pairs ? ((p > 0 ? 1 : 0) | (to_out ? 0 : 2)) : 0;
            unsigned int* ko = to_out ? reinterpret_cast<unsigned int*>(ranks_bev) : keys_t;
            unsigned int* vo = to_out ? reinterpret_cast<unsigned int*>(ranks_depth) : vals_t;
            if (p > 0) {
                FBBEV_LAUNCH(k_sort_hist_seg, wgs, FBBEV_SEG_NT, 0, stream, kin, (const int*)ctot, sg, p * srb, srb, skip, matrix,
                             (pm & 1) ? 2 : 1);
                FBBEV_CHECK_LAUNCH();
            }
            static const int swz = [] { const char* e = getenv("FBBEV_RANK_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();   // A/B knob, read once
            // XCD-contiguous chunk order when every XCD then owns whole samples' worth of chunks (B >= 8): measured 0.174 -> 0.156 ms
            // per build at BL2 B = 16, 0.192 -> 0.184 at the shipped grid B = 16; with few samples the round-robin order is faster
            // (BL2 B = 4: 0.123 vs 0.129) -- profiles/r04_time_rank_swizzle.jsonl
            const int sw = (swz && B >= 8) ? 1 : 0;
            FBBEV_LAUNCH(k_sort_scatter_seg, sw ? (wgs + 7) / 8 * 8 : wgs, FBBEV_SEG_NT, 0, stream, kin, vin, (const int*)matrix,
                         (const int*)ctot, sg, p, srb, skip, ko, vo, counts, sw, pm);
            FBBEV_CHECK_LAUNCH();
            kin = ko; vin = vo;
        }
    } else {
    if (cams) {
            fbbev_geom_src gs;
            gs.cam = *cams; gs.frustum = frustum; gs.gp = gp;
            FBBEV_SORT_DISPATCH(k_keys_hist_geom, L.v0, false, L.wgs0, stream, gs, n, rb, skip, keys_a, matrix);
        } else {
            FBBEV_SORT_DISPATCH(k_keys_hist_coor, L.v0, false, L.wgs0, stream, coor, n, n / B, gp, point_depth, depth_thr, rb, keys_a, matrix);
        }
        FBBEV_CHECK_LAUNCH();
        const unsigned int* kin = keys_a;
        const unsigned int* vin = nullptr;                         // pass 0: value = position = point id
        for (int p = 0; p < plan.passes; ++p) {
            const bool to_out = ((plan.passes - 1 - p) % 2) == 0;  // the last pass always writes the outputs
            unsigned int* ko = to_out ? reinterpret_cast<unsigned int*>(ranks_bev) : keys_t;
            unsigned int* vo = to_out ? reinterpret_cast<unsigned int*>(ranks_depth) : vals_t;
            const int v = p == 0 ? L.v0 : L.v1, wgs = p == 0 ? L.wgs0 : L.wgs1;
            if (p > 0) {                                           // count matrix of this pass (pass 0: written with the keys)
                FBBEV_SORT_DISPATCH(k_sort_hist, v, false, wgs, stream, kin, (const int*)counts, p * rb, rb, skip, matrix);
                FBBEV_CHECK_LAUNCH();
            }
            static const int swz = [] { const char* e = getenv("FBBEV_RANK_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();   // A/B knob, read once
            // pass 0 only: later passes size their grid for n but run on the P kept pairs -- a contiguous split would idle XCDs
            const bool sw = swz == 2 && p == 0;       // off by default: measured slower for the global sort (BL1 B = 1: 0.151 -> 0.156 ms)
            FBBEV_SORT_DISPATCH(k_sort_scatter, v, true, sw ? (wgs + 7) / 8 * 8 : wgs, stream, kin, vin, n, (const int*)matrix, p, rb, skip,
                                ko, vo, counts, sw ? wgs : 0);
            FBBEV_CHECK_LAUNCH();
            kin = ko; vin = vo;
        }
}
    const unsigned int* keys = reinterpret_cast<const unsigned int*>(ranks_bev);
    const unsigned int* vals = reinterpret_cast<const unsigned int*>(ranks_depth);
    const fbbev_fastdiv div_dhw = make_fastdiv((unsigned int)((long long)D * H * W)), div_hw = make_fastdiv((unsigned int)(H * W));
    switch (L.vi) {
        case 0:
            FBBEV_LAUNCH((k_interval_count<4, 4>), L.wgsi, 256, 0, stream, keys, (const int*)counts, skip, chunk_info);
            FBBEV_LAUNCH((k_interval_write<4, 4>), L.wgsi, 256, (size_t)2 * 4 * 256 * 4, stream, keys, vals, div_dhw, div_hw, (const int2*)chunk_info, skip,
                         ranks_feat, interval_starts, interval_lengths, interval_rank, counts);
            break;
        case 1:
            FBBEV_LAUNCH((k_interval_count<16, 4>), L.wgsi, 1024, 0, stream, keys, (const int*)counts, skip, chunk_info);
            FBBEV_LAUNCH((k_interval_write<16, 4>), L.wgsi, 1024, (size_t)2 * 4 * 1024 * 4, stream, keys, vals, div_dhw, div_hw, (const int2*)chunk_info, skip,
                         ranks_feat, interval_starts, interval_lengths, interval_rank, counts);
            break;
        default:
            FBBEV_LAUNCH((k_interval_count<16, 8>), L.wgsi, 1024, 0, stream, keys, (const int*)counts, skip, chunk_info);
            FBBEV_LAUNCH((k_interval_write<16, 8>), L.wgsi, 1024, (size_t)2 * 8 * 1024 * 4, stream, keys, vals, div_dhw, div_hw, (const int2*)chunk_info, skip,
                         ranks_feat, interval_starts, interval_lengths, interval_rank, counts);
            break;
    }
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_rank_build(const float* coor, int B, int N, int D, int H, int W,
                                const float* lower3, const float* interval3,
                                const float* grid_size3, int32_t* ranks_bev, int32_t* ranks_depth,
                                int32_t* ranks_feat, int32_t* interval_starts,
                                int32_t* interval_lengths, int32_t* interval_rank, int32_t* counts,
                                void* workspace, size_t workspace_bytes, fbbev_stream_t stream_) {
    if (!coor) return FBBEV_E_BADARG;
    return rank_build_impl(coor, nullptr, nullptr, B, N, D, H, W, lower3, interval3, grid_size3, ranks_bev, ranks_depth,
                           ranks_feat, interval_starts, interval_lengths, interval_rank, counts, workspace,
                           workspace_bytes, (fbbev_rt_stream)stream_);
}

extern "C" int fbbev_rank_build_depth(const float* coor, const float* depth, float depth_threshold, int B, int N, int D,
                                      int H, int W, const float* lower3, const float* interval3,
                                      const float* grid_size3, int32_t* ranks_bev, int32_t* ranks_depth,
                                      int32_t* ranks_feat, int32_t* interval_starts, int32_t* interval_lengths,
                                      int32_t* interval_rank, int32_t* counts, void* workspace, size_t workspace_bytes,
                                      fbbev_stream_t stream_) {
    if (!coor || !depth) return FBBEV_E_BADARG;
    return rank_build_impl(coor, nullptr, nullptr, B, N, D, H, W, lower3, interval3, grid_size3, ranks_bev, ranks_depth,
                           ranks_feat, interval_starts, interval_lengths, interval_rank, counts, workspace,
                           workspace_bytes, (fbbev_rt_stream)stream_, depth, depth_threshold);
}

static int lift_rank_build_impl(const float* frustum, const float* xs, const float* ys, const float* ds,
                                const float* rots, const float* trans, const float* intrins, const float* post_rots,
                                const float* post_trans, const float* bda, int B, int N, int D, int H, int W,
                                const float* lower3, const float* interval3, const float* grid_size3,
                                int32_t* ranks_bev, int32_t* ranks_depth, int32_t* ranks_feat, int32_t* interval_starts,
                                int32_t* interval_lengths, int32_t* interval_rank, int32_t* counts, void* workspace,
                                size_t workspace_bytes, fbbev_rt_stream stream, uint32_t* cam_key, int32_t* cache_state) {
    if (!xs || !ys || !ds || !rots || !trans || !intrins || !post_rots || !post_trans || !bda) return FBBEV_E_BADARG;
    if (B <= 0 || N <= 0) return FBBEV_E_BADARG;
    fbbev_cam_ptrs g;
    g.xs = xs; g.ys = ys; g.ds = ds; g.rots = rots; g.trans = trans; g.intrins = intrins; g.post_rots = post_rots;
    g.post_trans = post_trans; g.bda = bda; g.N = N; g.D = D; g.H = H; g.W = W;
    const int* skip = nullptr;
    if (cam_key) {
        if (!cache_state) return FBBEV_E_BADARG;
        FBBEV_LAUNCH(k_cam_key, 1, 256, 0, stream, g, B, cam_key, cache_state);
        FBBEV_CHECK_LAUNCH();
        skip = cache_state;
    }
    return rank_build_impl(nullptr, &g, frustum, B, N, D, H, W, lower3, interval3, grid_size3, ranks_bev, ranks_depth,
                           ranks_feat, interval_starts, interval_lengths, interval_rank, counts, workspace,
                           workspace_bytes, stream, nullptr, 0.f, skip);
}

extern "C" int fbbev_lift_rank_build(const float* frustum, const float* xs, const float* ys, const float* ds,
                                     const float* rots,
                                     const float* trans, const float* intrins, const float* post_rots,
                                     const float* post_trans, const float* bda, int B, int N, int D, int H,
                                     int W, const float* lower3, const float* interval3,
                                     const float* grid_size3, int32_t* ranks_bev, int32_t* ranks_depth,
                                     int32_t* ranks_feat, int32_t* interval_starts, int32_t* interval_lengths,
                                     int32_t* interval_rank, int32_t* counts, void* workspace,
                                     size_t workspace_bytes, fbbev_stream_t stream_) {
    return lift_rank_build_impl(frustum, xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, B, N, D, H, W, lower3,
                                interval3, grid_size3, ranks_bev, ranks_depth, ranks_feat, interval_starts,
                                interval_lengths, interval_rank, counts, workspace, workspace_bytes,
                                (fbbev_rt_stream)stream_, nullptr, nullptr);
}

extern "C" size_t fbbev_cam_key_words(int B, int N) {
    return (B <= 0 || N <= 0) ? 0 : (size_t)B * N * 33 + (size_t)B * 9;
}

extern "C" int fbbev_lift_rank_build_cached(const float* frustum, const float* xs, const float* ys, const float* ds,
                                            const float* rots, const float* trans, const float* intrins,
                                            const float* post_rots, const float* post_trans, const float* bda, int B,
                                            int N, int D, int H, int W, const float* lower3, const float* interval3,
                                            const float* grid_size3, int32_t* ranks_bev, int32_t* ranks_depth,
                                            int32_t* ranks_feat, int32_t* interval_starts, int32_t* interval_lengths,
                                            int32_t* interval_rank, int32_t* counts, void* workspace,
                                            size_t workspace_bytes, uint32_t* cam_key, int32_t* cache_state,
                                            fbbev_stream_t stream_) {
    if (!cam_key || !cache_state) return FBBEV_E_BADARG;
    return lift_rank_build_impl(frustum, xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, B, N, D, H, W, lower3,
                                interval3, grid_size3, ranks_bev, ranks_depth, ranks_feat, interval_starts,
                                interval_lengths, interval_rank, counts, workspace, workspace_bytes,
                                (fbbev_rt_stream)stream_, cam_key, cache_state);
}

// ------------------------------------------------------------------------------ fused dense fwd
static inline int pick_tile(int tile_voxels) {
    switch (tile_voxels) { case 64: case 128: case 256: case 512: case 1024: return tile_voxels; default: return 64; }
}
// channels-last small tiles: 8 / 16 / 32 voxels per workgroup (one 16-byte store per thread)
static inline int small_tile_shift(int tile_voxels) {
    switch (tile_voxels) { case 8: return 3; case 16: return 4; case 32: return 5; default: return 0; }
}

extern "C" size_t fbbev_pool_dense_workspace_bytes(int B, int Z, int Y, int X) {
    if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 256;
    const long long yx = (long long)Y * X;
    const long long tiles = ((long long)B * Z * yx + 7) / 8 + (long long)B * Z;   // smallest tile (8 voxels) = most tiles
    return align_up((size_t)(tiles + 2) * 8, 256);                // two ints per tile
}

static int pool_tile_index_impl(const int32_t* interval_rank, const int32_t* interval_starts,
                                const int32_t* counts, int n_intervals_max, int B, int Z, int Y,
                                int X, int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                                fbbev_stream_t stream_, const int32_t* skip);

extern "C" int fbbev_pool_tile_index(const int32_t* interval_rank, const int32_t* interval_starts,
                                     const int32_t* counts, int n_intervals_max, int B, int Z, int Y,
                                     int X, int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                                     fbbev_stream_t stream_) {
    return pool_tile_index_impl(interval_rank, interval_starts, counts, n_intervals_max, B, Z, Y, X, tile_voxels, flags,
                                tile_ws, tile_ws_bytes, stream_, nullptr);
}

extern "C" int fbbev_pool_tile_index_cached(const int32_t* interval_rank, const int32_t* interval_starts,
                                            const int32_t* counts, int n_intervals_max, int B, int Z, int Y,
                                            int X, int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                                            const int32_t* cache_state, int32_t* table_gate,
                                            fbbev_stream_t stream_) {
    if (!cache_state || !table_gate) return FBBEV_E_BADARG;
    if ((flags & FBBEV_POOL_CHANNELS_LAST) && small_tile_shift(tile_voxels)) return FBBEV_E_UNSUPPORTED;
    // keep THIS table only if the index set is unchanged and the table was built for this very build
    FBBEV_LAUNCH(k_tile_table_gate, 1, 1, 0, (fbbev_rt_stream)stream_, cache_state, table_gate);
    FBBEV_CHECK_LAUNCH();
    return pool_tile_index_impl(interval_rank, interval_starts, counts, n_intervals_max, B, Z, Y, X, tile_voxels, flags,
                                tile_ws, tile_ws_bytes, stream_, table_gate + 1);
}

static int pool_tile_index_impl(const int32_t* interval_rank, const int32_t* interval_starts,
                                const int32_t* counts, int n_intervals_max, int B, int Z, int Y,
                                int X, int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                                fbbev_stream_t stream_, const int32_t* skip) {
    if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0 || n_intervals_max < 0) return FBBEV_E_BADARG;
    if (!interval_rank || !interval_starts || !counts || !tile_ws) return FBBEV_E_BADARG;
    const long long yx = (long long)Y * X;
    if ((long long)B * Z * yx >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if ((flags & FBBEV_POOL_CHANNELS_LAST) && small_tile_shift(tile_voxels)) {
        // small tiles: scatter-built table (first/last interval per tile), -1 = empty
        const int sh = small_tile_shift(tile_voxels);
        const long long nvox = (long long)B * Z * yx;
        const long long nt = (nvox + (1 << sh) - 1) >> sh;
        if (tile_ws_bytes < (size_t)nt * 8) return FBBEV_E_WORKSPACE;
        int* first = static_cast<int*>(tile_ws);
        int e = fbbev_rt_memset_async(first, 0xFF, (size_t)nt * 8, (fbbev_rt_stream)stream_);
        if (e) return e;
        long long blocks = ((long long)n_intervals_max + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        FBBEV_LAUNCH(k_tile_scatter, blocks, 256, 0, (fbbev_rt_stream)stream_, interval_rank, counts,
                     n_intervals_max, sh, nvox, first, first + nt);
        FBBEV_CHECK_LAUNCH();
        return 0;
    }
    const int TV = pick_tile(tile_voxels);
    int tiles_per_plane = (int)((yx + TV - 1) / TV);
    long long n_tiles = (long long)B * Z * tiles_per_plane;
    long long plane = yx;
    if (flags & FBBEV_POOL_CHANNELS_LAST) {   // flat tiling of the whole (B*Z*Y*X) rank space
        plane = (long long)B * Z * yx;
        n_tiles = (plane + TV - 1) / TV;
        tiles_per_plane = (int)n_tiles + 1;   // every tile index t (incl. t == n_tiles) stays in "plane" 0
    }
    if (tile_ws_bytes < (size_t)(n_tiles + 1) * 8) return FBBEV_E_WORKSPACE;
    FBBEV_LAUNCH(k_tile_lower_bound2, (n_tiles + 1 + 255) / 256, 256, 0, (fbbev_rt_stream)stream_,
                 (int)n_tiles, tiles_per_plane, (int)plane, TV, interval_rank, interval_starts, counts,
                 n_intervals_max, skip, static_cast<int*>(tile_ws));
    FBBEV_CHECK_LAUNCH();
    return 0;
}

struct dense2_args {
    long long n_blocks; size_t lds; fbbev_rt_stream stream; int C, Z, yx, tpp, csplit, swizzle;
    long long stride_b, stride_c;
    const float *depth, *feat; const int32_t *rd, *rf, *irank, *starts, *lengths; const int* tile_meta;
    const float* addend = nullptr;   // optional (B,C,Y,X) broadcast-over-z add in the dense2 epilogue
    float* out;
    bool split = false;              // FBBEV_POOL_SPLIT_LONG: long intervals summed by the whole workgroup (tolerance mode)
    bool gather8 = false;            // FBBEV_POOL_GATHER8: eight points per gather batch (experiment knob, same bits)
};

#define FBBEV_POOL_SPLIT_LEN 32      // points above which an interval is split over the lane groups in tolerance mode
template <int TV, int CPL, int ST, int NT, int OT, bool T16 = false, int SPLIT = 0, int GU = 4>
static int launch_dense2(const dense2_args& a) {
    size_t lds = a.lds;
    if (SPLIT > 0) lds += ((size_t)TV + 4) * sizeof(int) + (size_t)FBBEV_POOL_SPLIT_GROUPS * (a.C / a.csplit) * sizeof(float);   // long-interval list + partial sums
    if (T16)                                   // 16-bit tile [CC][TV + 8] instead of fp32 [CC][TV + 4]
        lds = (size_t)(a.C / a.csplit) * (TV + 8) * 2 + ((size_t)3 * TV + 2 * FBBEV_NP_STAGE) * sizeof(int);
    if (lds > 64 * 1024) {  // > 64 KiB of dynamic LDS must be opted into (160 KiB per CU on gfx950)
        int e = fbbev_rt_allow_dyn_lds((const void*)k_pool_fwd_dense2<TV, CPL, ST, NT, OT, T16, 0, SPLIT, GU>, lds);
        if (e) return e;
    }
    long long grid = a.n_blocks;
    if (a.swizzle) {  // round up to whole groups of 8 chunks
        const long long g = 8ll << (a.swizzle - 1);
        grid = (a.n_blocks + g - 1) / g * g;
    }
    FBBEV_LAUNCH((k_pool_fwd_dense2<TV, CPL, ST, NT, OT, T16, 0, SPLIT, GU>), grid, NT, lds, a.stream, a.C, a.Z, a.yx, a.tpp,
                 a.csplit, (int)a.n_blocks, a.swizzle, a.stride_b, a.stride_c, a.depth, a.feat, a.rd, a.rf, a.irank, a.starts, a.lengths, a.tile_meta, a.addend, a.out);
    return fbbev_rt_last_error();
}

template <int TV, int CPL, int ST>
static int launch_dense2_nt(int nt, int ot, const dense2_args& a) {
    if (a.split) {                // tolerance mode: fp32 volume, default store policy, 256 threads, 64- / 128-voxel tiles
        if constexpr (ST == 4 && (TV == 64 || TV == 128)) {
            if (ot == 0 && nt == 256) return launch_dense2<TV, CPL, 4, 256, 0, false, FBBEV_POOL_SPLIT_LEN>(a);
        }
        return FBBEV_E_UNSUPPORTED;
    }
    if (a.gather8) {              // experiment knob: fp32 volume, default store policy, 256 threads, 64- / 128-voxel tiles
        if constexpr (ST == 4 && (TV == 64 || TV == 128)) {
            if (ot == 0 && nt == 256) return launch_dense2<TV, CPL, 4, 256, 0, false, 0, 8>(a);
        }
        return FBBEV_E_UNSUPPORTED;
    }
    if constexpr (ST == 4) {      // 16-bit output storage is built for the default store policy only
        if (ot != 0 && !a.addend && nt == 256)        // no epilogue add: the LDS tile itself is 16-bit (see the kernel)
            return ot == 1 ? launch_dense2<TV, CPL, 4, 256, 1, true>(a) : launch_dense2<TV, CPL, 4, 256, 2, true>(a);
        if (ot == 1) return nt == 128 ? launch_dense2<TV, CPL, 4, 128, 1>(a) : launch_dense2<TV, CPL, 4, 256, 1>(a);
        if (ot == 2) return nt == 128 ? launch_dense2<TV, CPL, 4, 128, 2>(a) : launch_dense2<TV, CPL, 4, 256, 2>(a);
    }
    return nt == 128 ? launch_dense2<TV, CPL, ST, 128, 0>(a) : launch_dense2<TV, CPL, ST, 256, 0>(a);
}

template <int TV, int CPL, int ST, int NT>
static int launch_dense_cl(const dense2_args& a) {
    long long grid = a.n_blocks;
    if (a.swizzle) {
        const long long g = 8ll << (a.swizzle - 1);
        grid = (a.n_blocks + g - 1) / g * g;
    }
    FBBEV_LAUNCH((k_pool_fwd_dense_cl<TV, CPL, ST, NT>), grid, NT, a.lds, a.stream, a.C, a.yx, (int)a.n_blocks,
                 a.swizzle, a.depth, a.feat, a.rd, a.rf, a.irank, a.starts, a.lengths, a.tile_meta, a.out);
    return fbbev_rt_last_error();
}

template <int TV, int CPL>
static int launch_dense_cl_st(int st, const dense2_args& a) {
    switch (st) {
        case 0: return launch_dense_cl<TV, CPL, 0, 256>(a);
        case 1: return launch_dense_cl<TV, CPL, 1, 256>(a);
        default: return launch_dense_cl<TV, CPL, 4, 256>(a);
    }
}

// store cache policy: 0 plain, 1 nontemporal, anything else -> `sc1 nt` (4), the measured best
template <int TV, int CPL>
static int launch_dense2_st(int st, int nt, int ot, const dense2_args& a) {
    if (ot != 0) return launch_dense2_nt<TV, CPL, 4>(nt, ot, a);
    switch (st) {
        case 0: return launch_dense2_nt<TV, CPL, 0>(nt, 0, a);
        case 1: return launch_dense2_nt<TV, CPL, 1>(nt, 0, a);
        default: return launch_dense2_nt<TV, CPL, 4>(nt, 0, a);
    }
}

static int pool_dense_fwd_impl(const float* depth, const float* feat,
                               const int32_t* ranks_depth, const int32_t* ranks_feat,
                               const int32_t* interval_rank, const int32_t* interval_starts,
                               const int32_t* interval_lengths, int B, int C, int Z, int Y,
                               int X, float* out, long long out_stride_b, long long out_stride_c,
                               const void* tile_ws, size_t tile_ws_bytes, int tile_voxels,
                               int flags, const float* addend, fbbev_stream_t stream_) {
    if (B <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0) return FBBEV_E_BADARG;
    if (!depth || !feat || !ranks_depth || !ranks_feat || !interval_rank || !interval_starts ||
        !interval_lengths || !out || !tile_ws) return FBBEV_E_BADARG;
    const long long yx = (long long)Y * X;
    if (C % 4 != 0 || C > 256 || yx % 4 != 0 || !aligned16(out) || !aligned16(feat)) return FBBEV_E_UNSUPPORTED;
    if ((long long)B * Z * yx >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (out_stride_c == 0) out_stride_c = (long long)Z * yx;            // contiguous (B,C,Z,Y,X)
    if (out_stride_b == 0) out_stride_b = (long long)C * out_stride_c;
    if (out_stride_c < (long long)Z * yx || out_stride_b < (long long)C * out_stride_c || out_stride_c % 4 != 0 ||
        out_stride_b % 4 != 0) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    const int TV = pick_tile(tile_voxels);
    const int tiles_per_plane = (int)((yx + TV - 1) / TV);
    const long long n_tiles = (long long)B * Z * tiles_per_plane;
    if (tile_ws_bytes < (size_t)(n_tiles + 1) * 8) return FBBEV_E_WORKSPACE;
    const int st = (flags & FBBEV_POOL_STORE_MASK) | ((flags >> FBBEV_POOL_STORE_HI_SHIFT) & 1) << 2;
    const int ot = (flags & FBBEV_POOL_OUT_BF16) ? 1 : ((flags & FBBEV_POOL_OUT_F16) ? 2 : 0);
    if (ot != 0) {   // 16-bit storage: 8 elements per 16-byte store
        if ((flags & FBBEV_POOL_OUT_BF16) && (flags & FBBEV_POOL_OUT_F16)) return FBBEV_E_BADARG;
        if (flags & FBBEV_POOL_CHANNELS_LAST) return FBBEV_E_UNSUPPORTED;
        if (yx % 8 != 0 || out_stride_c % 8 != 0 || out_stride_b % 8 != 0) return FBBEV_E_UNSUPPORTED;
    }
    if (addend && ((flags & FBBEV_POOL_CHANNELS_LAST) || !aligned16(addend))) return FBBEV_E_UNSUPPORTED;
    if (flags & FBBEV_POOL_CHANNELS_LAST) {
        // out is (B,Z,Y,X,C) contiguous: flat tiles, linear store stream, no LDS value tile
        if (out_stride_b != (long long)C * Z * yx || out_stride_c != (long long)Z * yx) return FBBEV_E_BADARG;
        const long long nvox = (long long)B * Z * yx;
        if (const int sh = small_tile_shift(tile_voxels)) {
            const long long nt = (nvox + (1 << sh) - 1) >> sh;
            if (tile_ws_bytes < (size_t)nt * 8) return FBBEV_E_WORKSPACE;
            int threads = (C / 4) << sh;
            if (threads > 1024) return FBBEV_E_UNSUPPORTED;
            threads = (threads + 63) / 64 * 64;
            int swz = 0;
            if (flags & FBBEV_POOL_XCD_SWIZZLE) {
                int lg = (flags >> FBBEV_POOL_SWZ_CHUNK_SHIFT) & 0x1F;
                swz = lg + 1;            // lg == 0 -> chunks of one tile (== plain round-robin)
            }
            long long grid = nt;
            if (swz) { const long long g = 8ll << (swz - 1); grid = (nt + g - 1) / g * g; }
            const int* first = static_cast<const int*>(tile_ws);
#define FBBEV_CLS(STV)                                                                                          \
    FBBEV_LAUNCH(k_pool_fwd_cl_small<STV>, grid, threads, 0, stream, C, sh, nvox, (int)nt, swz, depth, feat,    \
                 ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, first, first + nt, out)
            if (st == 0) { FBBEV_CLS(0); } else if (st == 1) { FBBEV_CLS(1); } else { FBBEV_CLS(4); }
#undef FBBEV_CLS
            return fbbev_rt_last_error();
        }
        const bool cpl8cl = (flags & FBBEV_POOL_CPL8) && (C % 8 == 0);
        if (256 / (C / (cpl8cl ? 8 : 4)) < 1) return FBBEV_E_UNSUPPORTED;
        dense2_args a;
        a.n_blocks = (nvox + TV - 1) / TV;
        if (tile_ws_bytes < (size_t)(a.n_blocks + 1) * 8) return FBBEV_E_WORKSPACE;
        a.lds = (3 * (size_t)TV + 2 * FBBEV_NP_STAGE) * sizeof(int);
        a.stream = stream; a.C = C; a.Z = Z; a.yx = (int)nvox; a.tpp = 0; a.csplit = 1;
        a.swizzle = 0;
        if (flags & FBBEV_POOL_XCD_SWIZZLE) {
            int lg = (flags >> FBBEV_POOL_SWZ_CHUNK_SHIFT) & 0x1F;
            if (lg == 0) lg = 4;
            a.swizzle = lg + 1;
        }
        a.depth = depth; a.feat = feat; a.rd = ranks_depth; a.rf = ranks_feat; a.irank = interval_rank;
        a.starts = interval_starts; a.lengths = interval_lengths; a.tile_meta = static_cast<const int*>(tile_ws);
        a.out = out; a.stride_b = 0; a.stride_c = 0;
        if (TV == 64) return cpl8cl ? launch_dense_cl_st<64, 8>(st, a) : launch_dense_cl_st<64, 4>(st, a);
        if (TV == 128) return cpl8cl ? launch_dense_cl_st<128, 8>(st, a) : launch_dense_cl_st<128, 4>(st, a);
        if (TV == 256) return cpl8cl ? launch_dense_cl_st<256, 8>(st, a) : launch_dense_cl_st<256, 4>(st, a);
        if (TV == 512) return cpl8cl ? launch_dense_cl_st<512, 8>(st, a) : launch_dense_cl_st<512, 4>(st, a);
        return cpl8cl ? launch_dense_cl_st<1024, 8>(st, a) : launch_dense_cl_st<1024, 4>(st, a);
    }
    int csplit = (flags >> FBBEV_POOL_CSPLIT_SHIFT) & 0xF;
    if (csplit == 0xF) csplit = 20;
    if (csplit < 1) csplit = 1;
    if (C % (4 * csplit) != 0) csplit = 1;
    const int CC = C / csplit;
    const bool cpl8 = (flags & FBBEV_POOL_CPL8) && (CC % 8 == 0);
    int nt = 256;
    if (((flags >> FBBEV_POOL_WG_SHIFT) & 0x3) == 1) nt = 128;
    if (nt / (CC / (cpl8 ? 8 : 4)) < 1) return FBBEV_E_UNSUPPORTED;
    dense2_args a;
    a.n_blocks = n_tiles * csplit;
    a.lds = ((size_t)CC * (TV + 4) + 3 * (size_t)TV + 2 * FBBEV_NP_STAGE) * sizeof(float);
    a.stream = stream; a.C = C; a.Z = Z; a.yx = (int)yx; a.tpp = tiles_per_plane; a.csplit = csplit;
    a.swizzle = 0;
    if (flags & FBBEV_POOL_XCD_SWIZZLE) {
        int lg = (flags >> FBBEV_POOL_SWZ_CHUNK_SHIFT) & 0x1F;   // log2(tiles per chunk); 0 -> default
        if (lg == 0) lg = 6;
        a.swizzle = lg + 1;
    }
    if (a.n_blocks + 8 >= (1ll << 31) || a.lds > 160 * 1024) return FBBEV_E_UNSUPPORTED;
    a.depth = depth; a.feat = feat; a.rd = ranks_depth; a.rf = ranks_feat; a.irank = interval_rank;
    a.starts = interval_starts; a.lengths = interval_lengths; a.tile_meta = static_cast<const int*>(tile_ws);
    a.out = out; a.stride_b = out_stride_b; a.stride_c = out_stride_c; a.addend = addend;
    a.split = (flags & FBBEV_POOL_SPLIT_LONG) != 0;
    a.gather8 = (flags & FBBEV_POOL_GATHER8) != 0 && !a.split;
    if ((flags & FBBEV_POOL_PIPE) && ot == 0 && !a.split && nt == 256 && st >= 2 && (TV == 64 || TV == 128 || TV == 256)) {
        // tiles per workgroup: enough to amortise the pipeline's fill, few enough to keep >= ~8 workgroups per CU's worth of runs
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: the tests switch the run length inside one process
        const int tpw_env = [] { const char* e = getenv("FBBEV_POOL_PIPE_TPW"); return e ? atoi(e) : 0; }();
#else
        static const int tpw_env = [] { const char* e = getenv("FBBEV_POOL_PIPE_TPW"); return e ? atoi(e) : 0; }();   // tuning knob, read once
#endif
        int tpw = tpw_env > 0 ? (tpw_env > 64 ? 64 : tpw_env) : 4;
        if (tpw_env <= 0) while (tpw > 1 && (n_tiles / tpw) * csplit < 2048) tpw >>= 1;
        const long long groups = (n_tiles + tpw - 1) / tpw;
        a.n_blocks = groups * csplit;
        a.lds = ((size_t)CC * (TV + 4) + 2 * (3 * (size_t)TV + 2 * FBBEV_NP_STAGE)) * sizeof(float);
        if (a.lds > 160 * 1024) return FBBEV_E_UNSUPPORTED;
        long long grid = a.n_blocks;
        if (a.swizzle) {
            const long long g = 8ll << (a.swizzle - 1);
            grid = (a.n_blocks + g - 1) / g * g;
        }
#define FBBEV_POOL_PIPE_LAUNCH(TV_, CPL_)                                                                                  \
        do {                                                                                                               \
            if (a.lds > 64 * 1024) { int e = fbbev_rt_allow_dyn_lds((const void*)k_pool_fwd_dense_pipe<TV_, CPL_, 4, 256>, a.lds); if (e) return e; } \
            FBBEV_LAUNCH((k_pool_fwd_dense_pipe<TV_, CPL_, 4, 256>), grid, 256, a.lds, a.stream, a.C, a.Z, a.yx, a.tpp, a.csplit, \
                         (int)a.n_blocks, a.swizzle, tpw, (int)n_tiles, a.stride_b, a.stride_c, a.depth, a.feat, a.rd, a.rf,    \
                         a.irank, a.starts, a.lengths, a.tile_meta, a.addend, a.out);                                      \
        } while (0)
        if (TV == 64) { if (cpl8) FBBEV_POOL_PIPE_LAUNCH(64, 8); else FBBEV_POOL_PIPE_LAUNCH(64, 4); }
        else if (TV == 128) { if (cpl8) FBBEV_POOL_PIPE_LAUNCH(128, 8); else FBBEV_POOL_PIPE_LAUNCH(128, 4); }
        else { if (cpl8) FBBEV_POOL_PIPE_LAUNCH(256, 8); else FBBEV_POOL_PIPE_LAUNCH(256, 4); }
#undef FBBEV_POOL_PIPE_LAUNCH
        return fbbev_rt_last_error();
    }
    if (TV == 64) return cpl8 ? launch_dense2_st<64, 8>(st, nt, ot, a) : launch_dense2_st<64, 4>(st, nt, ot, a);
    if (TV == 128) return cpl8 ? launch_dense2_st<128, 8>(st, nt, ot, a) : launch_dense2_st<128, 4>(st, nt, ot, a);
    if (TV == 256) return cpl8 ? launch_dense2_st<256, 8>(st, nt, ot, a) : launch_dense2_st<256, 4>(st, nt, ot, a);
    if (TV == 512) return cpl8 ? launch_dense2_st<512, 8>(st, nt, ot, a) : launch_dense2_st<512, 4>(st, nt, ot, a);
    return cpl8 ? launch_dense2_st<1024, 8>(st, nt, ot, a) : launch_dense2_st<1024, 4>(st, nt, ot, a);
}

extern "C" int fbbev_bev_pool_v2_dense_fwd(const float* depth, const float* feat,
                                           const int32_t* ranks_depth, const int32_t* ranks_feat,
                                           const int32_t* interval_rank, const int32_t* interval_starts,
                                           const int32_t* interval_lengths, int B, int C, int Z, int Y,
                                           int X, float* out, long long out_stride_b, long long out_stride_c,
                                           const void* tile_ws, size_t tile_ws_bytes, int tile_voxels,
                                           int flags, fbbev_stream_t stream_) {
    return pool_dense_fwd_impl(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C,
                               Z, Y, X, out, out_stride_b, out_stride_c, tile_ws, tile_ws_bytes, tile_voxels, flags, nullptr,
                               stream_);
}

// Measurement aid (bench.py `roofline.store_floor_ms` / `no_gather_ms`): the default fp32 instantiation of the dense kernel
// (128-voxel tiles, 8 channels per lane, 256 threads, `sc1 nt` stores) with its gathers compiled out -- same grid, tile
// walk and XCD order as the product launch with these flags.  mode 1: store pattern alone; mode 2: all but the depth /
// feature gathers.  Writes zeros to `out`.
extern "C" int fbbev_diag_pool_store_floor(const float* depth, const float* feat, const int32_t* ranks_depth,
                                           const int32_t* ranks_feat, const int32_t* interval_rank,
                                           const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C,
                                           int Z, int Y, int X, float* out, const void* tile_ws, size_t tile_ws_bytes,
                                           int tile_voxels, int flags, int mode, fbbev_stream_t stream_) {
    if (B <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0 || mode < 1 || mode > 3) return FBBEV_E_BADARG;
    if (!depth || !feat || !ranks_depth || !ranks_feat || !interval_rank || !interval_starts || !interval_lengths || !out ||
        !tile_ws) return FBBEV_E_BADARG;
    const long long yx = (long long)Y * X;
    const bool bf16 = (flags & FBBEV_POOL_OUT_BF16) != 0;          // round 5: the 16-bit-tile instantiation of the bf16-storage leg
    if (pick_tile(tile_voxels) != 128 || tile_voxels != 128 || yx % (bf16 ? 8 : 4) != 0 || !aligned16(out) ||
        (flags & (FBBEV_POOL_CHANNELS_LAST | FBBEV_POOL_OUT_F16)) || (long long)B * Z * yx >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int st = (flags & FBBEV_POOL_STORE_MASK) | ((flags >> FBBEV_POOL_STORE_HI_SHIFT) & 1) << 2;
    int csplit = (flags >> FBBEV_POOL_CSPLIT_SHIFT) & 0xF;
    if (csplit == 0xF) csplit = 20;
    if (csplit < 1 || C % (4 * csplit) != 0) csplit = 1;
    const int CC = C / csplit;
    if (!(flags & FBBEV_POOL_CPL8) || CC % 8 != 0 || st < 2 || ((flags >> FBBEV_POOL_WG_SHIFT) & 0x3) != 0 || 256 / (CC / 8) < 1)
        return FBBEV_E_UNSUPPORTED;
    const int tiles_per_plane = (int)((yx + 127) / 128);
    const long long n_tiles = (long long)B * Z * tiles_per_plane;
    if (tile_ws_bytes < (size_t)(n_tiles + 1) * 8) return FBBEV_E_WORKSPACE;
    dense2_args a;
    a.n_blocks = n_tiles * csplit;
    a.lds = bf16 ? (size_t)CC * (128 + 8) * 2 + ((size_t)3 * 128 + 2 * FBBEV_NP_STAGE) * sizeof(int)
                 : ((size_t)CC * (128 + 4) + 3 * (size_t)128 + 2 * FBBEV_NP_STAGE) * sizeof(float);
    a.stream = (fbbev_rt_stream)stream_; a.C = C; a.Z = Z; a.yx = (int)yx; a.tpp = tiles_per_plane; a.csplit = csplit;
    a.swizzle = 0;
    if (flags & FBBEV_POOL_XCD_SWIZZLE) {
        int lg = (flags >> FBBEV_POOL_SWZ_CHUNK_SHIFT) & 0x1F;
        if (lg == 0) lg = 6;
        a.swizzle = lg + 1;
    }
    if (a.n_blocks + 8 >= (1ll << 31) || a.lds > 64 * 1024) return FBBEV_E_UNSUPPORTED;
    a.depth = depth; a.feat = feat; a.rd = ranks_depth; a.rf = ranks_feat; a.irank = interval_rank;
    a.starts = interval_starts; a.lengths = interval_lengths; a.tile_meta = static_cast<const int*>(tile_ws);
    a.out = out; a.stride_b = (long long)C * Z * yx; a.stride_c = (long long)Z * yx; a.addend = nullptr;
    long long grid = a.n_blocks;
    if (a.swizzle) { const long long g = 8ll << (a.swizzle - 1); grid = (a.n_blocks + g - 1) / g * g; }
#define FBBEV_DIAG_DENSE(OT_, T16_, MODE_)                                                                               \
    FBBEV_LAUNCH((k_pool_fwd_dense2<128, 8, 4, 256, OT_, T16_, MODE_>), grid, 256, a.lds, a.stream, a.C, a.Z, a.yx, a.tpp, \
                 a.csplit, (int)a.n_blocks, a.swizzle, a.stride_b, a.stride_c, a.depth, a.feat, a.rd, a.rf, a.irank,      \
                 a.starts, a.lengths, a.tile_meta, a.addend, a.out)
    if (bf16) { if (mode == 1) FBBEV_DIAG_DENSE(1, true, 1); else if (mode == 2) FBBEV_DIAG_DENSE(1, true, 2); else FBBEV_DIAG_DENSE(1, true, 3); }
    else { if (mode == 1) FBBEV_DIAG_DENSE(0, false, 1); else if (mode == 2) FBBEV_DIAG_DENSE(0, false, 2); else FBBEV_DIAG_DENSE(0, false, 3); }
#undef FBBEV_DIAG_DENSE
    return fbbev_rt_last_error();
}

extern "C" int fbbev_bev_pool_v2_dense_fwd_add(const float* depth, const float* feat,
                                               const int32_t* ranks_depth, const int32_t* ranks_feat,
                                               const int32_t* interval_rank, const int32_t* interval_starts,
                                               const int32_t* interval_lengths, int B, int C, int Z, int Y,
                                               int X, float* out, long long out_stride_b, long long out_stride_c,
                                               const void* tile_ws, size_t tile_ws_bytes, int tile_voxels,
                                               int flags, const float* addend, fbbev_stream_t stream_) {
    if (!addend) return FBBEV_E_BADARG;
    return pool_dense_fwd_impl(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C,
                               Z, Y, X, out, out_stride_b, out_stride_c, tile_ws, tile_ws_bytes, tile_voxels, flags, addend,
                               stream_);
}

// ------------------------------------------------------------------------------ lift-splat in ONE entry
// Workspace layout of fbbev_lift_splat_fused (all blocks 256-byte aligned): the seven padded index tensors + counts the build
// writes, the NHWC feature rows, the tile table, the ranking workspace.
struct lift_splat_layout {
    size_t ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths, interval_rank, counts, feat, tile, rank, total;
    size_t tile_bytes, rank_bytes;
};
static bool lift_splat_plan(int B, int N, int D, int H, int W, int C, int Z, int Y, int X, lift_splat_layout& L) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0) return false;
    const long long n = (long long)B * N * D * H * W;
    if (n >= (1ll << 31)) return false;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.ranks_bev = take((size_t)n * 4); L.ranks_depth = take((size_t)n * 4); L.ranks_feat = take((size_t)n * 4);
    L.interval_starts = take((size_t)n * 4); L.interval_lengths = take((size_t)n * 4); L.interval_rank = take((size_t)n * 4);
    L.counts = take(16);
    L.feat = take((size_t)B * N * H * W * C * 4);
    L.tile_bytes = fbbev_pool_dense_workspace_bytes(B, Z, Y, X);
    L.tile = take(L.tile_bytes);
    L.rank_bytes = fbbev_rank_workspace_bytes(n);
    L.rank = take(L.rank_bytes);
    L.total = off;
    return true;
}

extern "C" size_t fbbev_lift_splat_fused_ws_bytes(int B, int N, int D, int H, int W, int C, int Z, int Y, int X) {
    lift_splat_layout L;
    return lift_splat_plan(B, N, D, H, W, C, Z, Y, X, L) ? L.total : 0;
}

extern "C" int fbbev_lift_splat_fused_ws_offsets(int B, int N, int D, int H, int W, int C, int Z, int Y, int X, size_t* offsets8) {
    lift_splat_layout L;
    if (!offsets8 || !lift_splat_plan(B, N, D, H, W, C, Z, Y, X, L)) return FBBEV_E_BADARG;
    offsets8[0] = L.ranks_bev; offsets8[1] = L.ranks_depth; offsets8[2] = L.ranks_feat; offsets8[3] = L.interval_starts;
    offsets8[4] = L.interval_lengths; offsets8[5] = L.interval_rank; offsets8[6] = L.counts; offsets8[7] = L.feat;
    return 0;
}

extern "C" int fbbev_lift_splat_fused(const float* frustum, const float* xs, const float* ys, const float* ds, const float* rots,
                                      const float* trans, const float* intrins, const float* post_rots, const float* post_trans,
                                      const float* bda, const float* depth, const float* context, int B, int N, int D, int H,
                                      int W, int C, const float* lower3, const float* interval3, const float* grid_size3, int Z,
                                      int Y, int X, void* out, long long out_stride_b, long long out_stride_c, int tile_voxels,
                                      int flags, void* workspace, size_t workspace_bytes, uint32_t* cam_key, int32_t* cache_state,
                                      fbbev_stream_t stream_) {
    lift_splat_layout L;
    if (!depth || !context || !out || !workspace) return FBBEV_E_BADARG;
    if (!lift_splat_plan(B, N, D, H, W, C, Z, Y, X, L)) return FBBEV_E_BADARG;
    if ((cam_key == nullptr) != (cache_state == nullptr)) return FBBEV_E_BADARG;
    if (workspace_bytes < L.total) return FBBEV_E_WORKSPACE;
    if (!aligned16(workspace)) return FBBEV_E_UNSUPPORTED;
    char* ws = static_cast<char*>(workspace);
    auto i32 = [&](size_t o) { return reinterpret_cast<int32_t*>(ws + o); };
    const long long n = (long long)B * N * D * H * W;
    // view_transformer.py:458-498 + :547-605: geometry + voxel ranking, device-side counts (cache_state[0] = 1 on a key hit)
    int rc = lift_rank_build_impl(frustum, xs, ys, ds, rots, trans, intrins, post_rots, post_trans, bda, B, N, D, H, W, lower3,
                                  interval3, grid_size3, i32(L.ranks_bev), i32(L.ranks_depth), i32(L.ranks_feat),
                                  i32(L.interval_starts), i32(L.interval_lengths), i32(L.interval_rank), i32(L.counts),
                                  ws + L.rank, L.rank_bytes, (fbbev_rt_stream)stream_, cam_key, cache_state);
    if (rc) return rc;
    // :536 / bev_pool.py:18: feat.permute(0,1,3,4,2).contiguous()
    float* feat = reinterpret_cast<float*>(ws + L.feat);
    rc = fbbev_nchw_to_nhwc(context, feat, B * N, C, H * W, stream_);
    if (rc) return rc;
    // bev_pool.py:24-35,88: new_zeros + kernel + permute().contiguous() as tile index + one dense pass
    if (cache_state)
        rc = fbbev_pool_tile_index_cached(i32(L.interval_rank), i32(L.interval_starts), i32(L.counts), (int)n, B, Z, Y, X,
                                          tile_voxels, flags, ws + L.tile, L.tile_bytes, cache_state, cache_state + 2, stream_);
    else
        rc = pool_tile_index_impl(i32(L.interval_rank), i32(L.interval_starts), i32(L.counts), (int)n, B, Z, Y, X, tile_voxels,
                                  flags, ws + L.tile, L.tile_bytes, stream_, nullptr);
    if (rc) return rc;
    return pool_dense_fwd_impl(depth, feat, i32(L.ranks_depth), i32(L.ranks_feat), i32(L.interval_rank), i32(L.interval_starts),
                               i32(L.interval_lengths), B, C, Z, Y, X, static_cast<float*>(out), out_stride_b, out_stride_c,
                               ws + L.tile, L.tile_bytes, tile_voxels, flags, nullptr, stream_);
}

static int pool_zmean_impl(const float* depth, const float* feat, const int32_t* ranks_depth,
                           const int32_t* ranks_feat, const int32_t* interval_rank,
                           const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C, int Z,
                           int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                           int tile_voxels, int flags, int z_groups, float* partial, size_t partial_bytes, fbbev_stream_t stream_,
                           int rows_out = 0, const float* row_bias = nullptr);

extern "C" int fbbev_pool_zmean(const float* depth, const float* feat, const int32_t* ranks_depth,
                                const int32_t* ranks_feat, const int32_t* interval_rank,
                                const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C, int Z,
                                int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                                int tile_voxels, int flags, fbbev_stream_t stream_) {
    return pool_zmean_impl(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
                           out_mean, tile_ws, tile_ws_bytes, tile_voxels, flags, 1, nullptr, 0, stream_);
}

// fbbev_pool_zmean with the Z planes of a tile dealt to z_groups workgroups (each walks ceil(Z / z_groups) planes) and a caller-owned
// partial buffer of z_groups * B*C*Y*X floats; a second small kernel adds the groups in order and divides by Z.  For grids with few
// tiles (the shipped 100x100x8 grid): the single pass is one Z-plane latency chain per workgroup.
extern "C" int fbbev_pool_zmean_split(const float* depth, const float* feat, const int32_t* ranks_depth,
                                      const int32_t* ranks_feat, const int32_t* interval_rank,
                                      const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C, int Z,
                                      int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                                      int tile_voxels, int flags, int z_groups, void* partial_ws, size_t partial_ws_bytes,
                                      fbbev_stream_t stream_) {
    if (z_groups < 1 || z_groups > 64) return FBBEV_E_BADARG;
    if (z_groups > 1 && (!partial_ws || !aligned16(partial_ws))) return FBBEV_E_BADARG;
    return pool_zmean_impl(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
                           out_mean, tile_ws, tile_ws_bytes, tile_voxels, flags, z_groups, static_cast<float*>(partial_ws),
                           partial_ws_bytes, stream_);
}

static int pool_zmean_impl(const float* depth, const float* feat, const int32_t* ranks_depth,
                           const int32_t* ranks_feat, const int32_t* interval_rank,
                           const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C, int Z,
                           int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                           int tile_voxels, int flags, int z_groups, float* partial, size_t partial_bytes, fbbev_stream_t stream_,
                           int rows_out, const float* row_bias) {
    if (B <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0) return FBBEV_E_BADARG;
    if (rows_out && (z_groups != 1 || (row_bias && !aligned16(row_bias)))) return FBBEV_E_UNSUPPORTED;
    if (!depth || !feat || !ranks_depth || !ranks_feat || !interval_rank || !interval_starts || !interval_lengths ||
        !out_mean || !tile_ws) return FBBEV_E_BADARG;
    const long long yx = (long long)Y * X;
    if (C % 4 != 0 || C > 256 || yx % 4 != 0 || !aligned16(out_mean) || !aligned16(feat)) return FBBEV_E_UNSUPPORTED;
    if ((long long)B * Z * yx >= (1ll << 31) || (flags & FBBEV_POOL_CHANNELS_LAST)) return FBBEV_E_UNSUPPORTED;
    const int TV = pick_tile(tile_voxels);
    if (TV > 256) return FBBEV_E_UNSUPPORTED;
    const int tiles_per_plane = (int)((yx + TV - 1) / TV);
    const long long n_tiles = (long long)B * Z * tiles_per_plane;
    if (tile_ws_bytes < (size_t)(n_tiles + 1) * 8) return FBBEV_E_WORKSPACE;
    int csplit = (flags >> FBBEV_POOL_CSPLIT_SHIFT) & 0xF;
    if (csplit == 0xF) csplit = 20;
    if (csplit < 1) csplit = 1;
    if (C % (4 * csplit) != 0) csplit = 1;
    const int CC = C / csplit;
    const bool cpl8 = (flags & FBBEV_POOL_CPL8) && (CC % 8 == 0);
    if (256 / (CC / (cpl8 ? 8 : 4)) < 1) return FBBEV_E_UNSUPPORTED;
    const size_t lds = ((size_t)CC * (TV + 4) + 3 * (size_t)TV + 2 * FBBEV_NP_STAGE) * sizeof(float);
    if (lds > 64 * 1024) return FBBEV_E_UNSUPPORTED;
    const long long blocks = (long long)B * tiles_per_plane * csplit;
    if (z_groups > Z) z_groups = Z;
    const long long n_out = (long long)B * C * yx;
    if (z_groups > 1 && partial_bytes < (size_t)z_groups * n_out * sizeof(float)) return FBBEV_E_WORKSPACE;
    if (blocks * z_groups >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int* meta = static_cast<const int*>(tile_ws);
    // round 5: the column form (every plane's metadata at once, a lane group per pixel; the same bits) when the column's slot table
    // fits -- OPT-IN (FBBEV_ZMEAN_COL=1): built to cut the Z dependent round-trip chains of the plane walk, measured SLOWER at the
    // BASELINE configs[2] grid, B = 4 (135.7 vs 103.9 us, S3 1.395 vs 1.360 ms: profiles/r05_exp_zmean_col.md -- 54 KB of LDS = 2
    // workgroups per CU instead of 5, and a lane group's pixels are a serial chain of their own)
#ifdef FBBEV_TEST_OVERRIDES
    const bool col_on = [] { const char* e = getenv("FBBEV_ZMEAN_COL"); return e && atoi(e) != 0; }();
#else
    static const bool col_on = [] { const char* e = getenv("FBBEV_ZMEAN_COL"); return e && atoi(e) != 0; }();   // read once
#endif
    const size_t lds_col = fbbev_zmean_col_lds_bytes(CC, TV, Z);
    if (col_on && !rows_out && z_groups == 1 && Z <= 64 && lds_col <= 64 * 1024) {
#define FBBEV_ZMEAN_COL(TV_, CPL_)                                                                                   \
    FBBEV_LAUNCH((k_pool_zmean_col<TV_, CPL_, 256>), blocks, 256, lds_col, (fbbev_rt_stream)stream_, C, Z, (int)yx,   \
                 tiles_per_plane, csplit, (int)blocks, depth, feat, ranks_depth, ranks_feat, interval_rank,          \
                 interval_starts, interval_lengths, meta, out_mean)
        if (TV == 64) { if (cpl8) FBBEV_ZMEAN_COL(64, 8); else FBBEV_ZMEAN_COL(64, 4); }
        else if (TV == 128) { if (cpl8) FBBEV_ZMEAN_COL(128, 8); else FBBEV_ZMEAN_COL(128, 4); }
        else { if (cpl8) FBBEV_ZMEAN_COL(256, 8); else FBBEV_ZMEAN_COL(256, 4); }
#undef FBBEV_ZMEAN_COL
        FBBEV_CHECK_LAUNCH();
        return 0;
    }
#define FBBEV_ZMEAN(TV_, CPL_)                                                                                       \
    FBBEV_LAUNCH((k_pool_zmean<TV_, CPL_, 256>), blocks * z_groups, 256, lds, (fbbev_rt_stream)stream_, C, Z, (int)yx, \
                 tiles_per_plane, csplit, (int)blocks, depth, feat, ranks_depth, ranks_feat, interval_rank,          \
                 interval_starts, interval_lengths, meta, out_mean, z_groups, partial, rows_out, row_bias)
    if (TV == 64) { if (cpl8) FBBEV_ZMEAN(64, 8); else FBBEV_ZMEAN(64, 4); }
    else if (TV == 128) { if (cpl8) FBBEV_ZMEAN(128, 8); else FBBEV_ZMEAN(128, 4); }
    else { if (cpl8) FBBEV_ZMEAN(256, 8); else FBBEV_ZMEAN(256, 4); }
#undef FBBEV_ZMEAN
    FBBEV_CHECK_LAUNCH();
    if (z_groups > 1) {
        FBBEV_LAUNCH(k_pool_zmean_reduce, (n_out / 4 + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, (const float*)partial, n_out,
                     z_groups, (float)Z, out_mean);
        FBBEV_CHECK_LAUNCH();
    }
    return 0;
}

// fbbev_pool_zmean with the result written as the backward projection's QUERY ROWS: out_rows (B, Y*X, C) = mean + row_bias (Y*X, C)
// (bev_embedding; may be null) -- backward_projection.py:96-99's flatten + permute + `+ bev_embedding` done by the Z-mean's store
// instead of a transposing pass over (B, C, Y, X).  Single pass only (no Z groups).
extern "C" int fbbev_pool_zmean_rows(const float* depth, const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                                     const int32_t* interval_rank, const int32_t* interval_starts,
                                     const int32_t* interval_lengths, int B, int C, int Z, int Y, int X, const float* row_bias,
                                     float* out_rows, const void* tile_ws, size_t tile_ws_bytes, int tile_voxels, int flags,
                                     fbbev_stream_t stream_) {
    return pool_zmean_impl(depth, feat, ranks_depth, ranks_feat, interval_rank, interval_starts, interval_lengths, B, C, Z, Y, X,
                           out_rows, tile_ws, tile_ws_bytes, tile_voxels, flags, 1, nullptr, 0, stream_, 1, row_bias);
}

// ------------------------------------------------------------------------------ MSDeformAttn
extern "C" int fbbev_msda_fwd(const float* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* sampling_loc,
                              const float* attn_weight, int batch, int spatial_size, int num_heads,
                              int channels, int num_levels, int num_query, int num_point, float* out,
                              fbbev_stream_t stream_) {
    if (batch < 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 ||
        num_query < 0 || num_point <= 0) return FBBEV_E_BADARG;
    const long long n = (long long)batch * num_query * num_heads * channels;
    if (n == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out)
        return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FBBEV_LAUNCH(k_msda_fwd, blocks, 256, 0, stream, n, value, spatial_shapes, level_start_index,
                 sampling_loc, attn_weight, spatial_size, num_heads, channels, num_levels, num_query,
                 num_point, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_msda_fwd_fused(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* ref_points, const float* offsets, const float* attn_weight, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                    int num_point, int head_stride, int offsets_head_minor, float* out,
                                    fbbev_stream_t stream_) {
    if (batch < 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query < 0 ||
        num_point <= 0) return FBBEV_E_BADARG;
    const long long units = (long long)batch * num_query * num_heads;
    if (units == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !ref_points || !offsets || !attn_weight || !out)
        return FBBEV_E_BADARG;
    const int HS = head_stride == 0 ? channels : head_stride;
    if (HS < channels) return FBBEV_E_BADARG;
    if ((((uintptr_t)offsets) & 7) != 0) return FBBEV_E_UNSUPPORTED;
    const bool wide = HS % 4 == 0 && HS >= (channels + 3) / 4 * 4 && aligned16(value);
    const bool qi = (offsets_head_minor & 4) != 0;                 // value rows stored [chunk][head][4 floats]
    if (qi && !wide) return FBBEV_E_UNSUPPORTED;
    if ((long long)batch * spatial_size * num_heads * HS * 4 >= (1ll << 32)) return FBBEV_E_UNSUPPORTED;   // 32-bit byte offsets
    long long ub = ((units + 255) / 256 + 7) / 8 * 8;             // XCD-contiguous order: a multiple of 8 workgroups
    if (ub > 65536) ub = 65536;
    // attention weights staged through LDS ([256][L*P+1] floats) when they fit beside 4 workgroups per CU
    const int LP = num_levels * num_point;
    const bool stage = LP % 4 == 0 && LP <= 36 && aligned16(attn_weight);
    const size_t lds = stage ? (size_t)256 * (LP + 1) * sizeof(float) : 0;
#define FBBEV_MSDA_UNIT_W(DH_, W_, Q_)                                                                               \
    FBBEV_LAUNCH((k_msda_fwd_unit<DH_, W_, Q_>), ub, 256, lds, (fbbev_rt_stream)stream_, units, value,                \
                 spatial_shapes, level_start_index, ref_points, offsets, attn_weight, spatial_size, num_heads,        \
                 num_levels, num_query, num_point, HS, (offsets_head_minor & 1) ? 1 : 0, stage ? 1 : 0, out)
#define FBBEV_MSDA_UNIT(DH_) do { if (qi) FBBEV_MSDA_UNIT_W(DH_, true, true); else if (wide) FBBEV_MSDA_UNIT_W(DH_, true, false); \
                                  else FBBEV_MSDA_UNIT_W(DH_, false, false); } while (0)
    if (channels == 10) FBBEV_MSDA_UNIT(10);
    else if (channels == 8) FBBEV_MSDA_UNIT(8);
    else if (channels == 16) FBBEV_MSDA_UNIT(16);
    else if (channels == 32) FBBEV_MSDA_UNIT(32);
    else if (channels == 4) FBBEV_MSDA_UNIT(4);
    else return FBBEV_E_UNSUPPORTED;
#undef FBBEV_MSDA_UNIT
#undef FBBEV_MSDA_UNIT_W
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_msda_bwd(const float* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* sampling_loc,
                              const float* attn_weight, const float* grad_output, int batch,
                              int spatial_size, int num_heads, int channels, int num_levels,
                              int num_query, int num_point, float* grad_value,
                              float* grad_sampling_loc, float* grad_attn_weight,
                              fbbev_stream_t stream_) {
    if (batch < 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 ||
        num_query < 0 || num_point <= 0) return FBBEV_E_BADARG;
    const long long n_units = (long long)batch * num_query * num_heads;
    if (n_units == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight ||
        !grad_output || !grad_value || !grad_sampling_loc || !grad_attn_weight) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
#define FBBEV_MSDA_BWD(GW)                                                                          \
    FBBEV_LAUNCH(k_msda_bwd<GW>, (n_units * GW + 255) / 256, 256, 0, stream, n_units, value,       \
                 spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,         \
                 spatial_size, num_heads, channels, num_levels, num_query, num_point, grad_value,   \
                 grad_sampling_loc, grad_attn_weight)
    if (channels <= 16) FBBEV_MSDA_BWD(16);
    else if (channels <= 32) FBBEV_MSDA_BWD(32);
    else FBBEV_MSDA_BWD(64);
#undef FBBEV_MSDA_BWD
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// Band-binned fixed-point backward (msda_bwd_kernels.h).  level_hw_host: HOST array of L (h, w) pairs -- the band count is a
// launch dimension.  Plan: tokens per LDS plane from the LDS budget (64 KB: two workgroups per CU; FBBEV_MSDA_BWD_LDS_KB,
// read once, tunes it), a band = whole rows of one level.
struct msda_bwd_plan { int budget, n_bands; size_t lds, ws_ranges, ws; };
static bool msda_bwd_plan_for(int B, int S, int M, int Dh, int L, int Q, int P, const int32_t* level_hw, msda_bwd_plan* pl) {
    if (!level_hw || !(Dh == 4 || Dh == 8 || Dh == 10 || Dh == 16 || Dh == 32) || Q <= 0) return false;
    // measured at BASELINE configs[2] (200 x 200 BEV, B = 4): scatter 2.24 / 0.80 / 0.45 ms with 32 / 64 / 128 KB planes -- the
    // halo of re-evaluated queries per band is what costs, so a band is as tall as LDS allows (one workgroup per CU) ...
    auto read_kb = [] { const char* e = getenv("FBBEV_MSDA_BWD_LDS_KB"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 150 ? v : 144; };
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: the tests switch budgets inside one process
    const int lds_kb = read_kb();
#else
    static const int lds_kb = read_kb();          // read once: ws_bytes() and the launch must agree
#endif
    int budget = (lds_kb * 1024) / (Dh * (int)sizeof(long long)) - 16;
    long long nb = 0, tokens = 0;
    int max_tok = 0, max_w = 1;
    for (int l = 0; l < L; ++l) max_w = level_hw[2 * l + 1] > max_w ? level_hw[2 * l + 1] : max_w;
    for (;;) {
        nb = 0; tokens = 0; max_tok = 0;
        for (int l = 0; l < L; ++l) {
            const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
            if (h <= 0 || w <= 0 || h > 32767 || w > budget) return false;
            const int rpb = budget / w;
            nb += (h + rpb - 1) / rpb;
            tokens += (long long)h * w;
            const int t = (rpb < h ? rpb : h) * w;
            max_tok = t > max_tok ? t : max_tok;
        }
        // ... unless that leaves CUs idle: shrink the bands until there is a workgroup per CU (small grids / small batches)
        if ((long long)B * M * nb >= 256 || budget * 3 / 4 < 2 * max_w) break;
        budget = budget * 3 / 4;
    }
    if (tokens != S || nb > 65535 || (long long)B * M * nb >= (1ll << 31)) return false;
    pl->budget = budget;
    pl->n_bands = (int)nb;
    pl->lds = (size_t)FBBEV_DA_PLANE_WORDS(max_tok, Dh) * sizeof(long long);
    pl->ws_ranges = ((size_t)B * L * Q * sizeof(unsigned int) + 255) / 256 * 256;
    pl->ws = pl->ws_ranges + (size_t)B * nb * 2 * sizeof(int);
    return true;
}

extern "C" size_t fbbev_msda_bwd_ws_bytes(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                          int num_query, int num_point, const int32_t* level_hw_host) {
    msda_bwd_plan pl;
    if (batch <= 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_point <= 0 ||
        !msda_bwd_plan_for(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, level_hw_host, &pl))
        return 0;
    return pl.ws;
}

extern "C" int fbbev_msda_bwd_ws(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const float* sampling_loc, const float* attn_weight, const float* grad_output, int batch,
                                 int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                                 float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                 const int32_t* level_hw_host, void* ws, size_t ws_bytes, fbbev_stream_t stream_) {
    if (batch < 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query < 0 ||
        num_point <= 0) return FBBEV_E_BADARG;
    msda_bwd_plan pl;
    if (batch == 0 || num_query == 0 || !ws ||
        !msda_bwd_plan_for(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, level_hw_host, &pl) ||
        ws_bytes < pl.ws)
        return fbbev_msda_bwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, batch,
                              spatial_size, num_heads, channels, num_levels, num_query, num_point, grad_value,
                              grad_sampling_loc, grad_attn_weight, stream_);
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_output || !grad_value ||
        !grad_sampling_loc || !grad_attn_weight) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    const int B = batch, S = spatial_size, M = num_heads, Dh = channels, L = num_levels, Q = num_query, P = num_point;
    unsigned int* ranges = static_cast<unsigned int*>(ws);
    int* qrange = reinterpret_cast<int*>(static_cast<char*>(ws) + pl.ws_ranges);
    {
        const long long n = (long long)B * L * Q;
        long long blocks = (n + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        FBBEV_LAUNCH(k_msda_row_ranges, blocks, 256, 0, stream, n, spatial_shapes, sampling_loc, M, L, Q, P, ranges);
        FBBEV_CHECK_LAUNCH();
        FBBEV_LAUNCH(k_msda_band_queries, (long long)B * pl.n_bands, 256, 0, stream, spatial_shapes, level_start_index,
                     (const unsigned int*)ranges, L, Q, pl.n_bands, pl.budget, qrange);
        FBBEV_CHECK_LAUNCH();
    }
    {
        // unit-owned gradients: one lane per unit with 8-byte corner loads when the rows allow it, else the atomic kernel's
        // lane groups without its atomics
        const long long n_units = (long long)B * Q * M;
        const bool lane_units = Dh % 2 == 0 && ((uintptr_t)value & 7) == 0 && ((uintptr_t)grad_output & 7) == 0 &&
                                ((uintptr_t)sampling_loc & 7) == 0 && ((uintptr_t)grad_sampling_loc & 7) == 0 &&      // 8-byte location / gradient pairs (round 6)
                                (n_units + 255) / 256 < (1ll << 31);
#define FBBEV_MSDA_BWD_UL(DH_)                                                                                        \
    FBBEV_LAUNCH((k_msda_bwd_unit<DH_>), (n_units + 255) / 256, 256, 0, stream, n_units, value, spatial_shapes,         \
                 level_start_index, sampling_loc, attn_weight, grad_output, S, M, L, Q, P, grad_sampling_loc, grad_attn_weight)
#define FBBEV_MSDA_BWD_U(GW)                                                                                          \
    FBBEV_LAUNCH((k_msda_bwd<GW, false>), (n_units * GW + 255) / 256, 256, 0, stream, n_units, value, spatial_shapes, \
                 level_start_index, sampling_loc, attn_weight, grad_output, S, M, Dh, L, Q, P, grad_value,             \
                 grad_sampling_loc, grad_attn_weight)
        if (lane_units && Dh == 10) FBBEV_MSDA_BWD_UL(10);
        else if (lane_units && Dh == 8) FBBEV_MSDA_BWD_UL(8);
        else if (lane_units && Dh == 4) FBBEV_MSDA_BWD_UL(4);
        else if (lane_units && Dh == 16) FBBEV_MSDA_BWD_UL(16);
        else if (Dh <= 16) FBBEV_MSDA_BWD_U(16);
        else FBBEV_MSDA_BWD_U(32);
#undef FBBEV_MSDA_BWD_U
#undef FBBEV_MSDA_BWD_UL
        FBBEV_CHECK_LAUNCH();
    }
    const long long wgs = (long long)B * M * pl.n_bands;
#define FBBEV_MSDA_BWD_SC(DH_)                                                                                        \
    do {                                                                                                              \
        int e_ = fbbev_rt_allow_dyn_lds((const void*)k_msda_bwd_scatter<512, DH_>, pl.lds);                            \
        if (e_) return e_;                                                                                            \
        FBBEV_LAUNCH((k_msda_bwd_scatter<512, DH_>), wgs, 512, pl.lds, stream, spatial_shapes, level_start_index,      \
                     sampling_loc, attn_weight, grad_output, (const unsigned int*)ranges, (const int*)qrange, S, M, L,  \
                     Q, P, pl.n_bands, pl.budget, grad_value);                                                        \
    } while (0)
    if (Dh == 10) FBBEV_MSDA_BWD_SC(10);
    else if (Dh == 8) FBBEV_MSDA_BWD_SC(8);
    else if (Dh == 4) FBBEV_MSDA_BWD_SC(4);
    else if (Dh == 16) FBBEV_MSDA_BWD_SC(16);
    else FBBEV_MSDA_BWD_SC(32);
#undef FBBEV_MSDA_BWD_SC
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ fused DA cross-attention
// 16-bit token rows (value_elem_type 1 = bf16, 2 = f16): chunk-major rows of 8-element pieces, unit-per-lane kernel only
extern "C" int fbbev_da_cross_attn_fwd_e(const void* value, const int64_t* spatial_shapes,
                                         const int64_t* level_start_index, const float* pred_depth,
                                         const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                         const float* offsets, const float* attn, int B, int Ncam, int S, int M,
                                         int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                         int head_minor, int head_stride, int value_elem_type, float* slots,
                                         fbbev_stream_t stream_) {
    if (value_elem_type == 0)
        return fbbev_da_cross_attn_fwd(static_cast<const float*>(value), spatial_shapes, level_start_index, pred_depth, ref_cam,
                                       mask, qdepth, offsets, attn, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor,
                                       head_stride, slots, stream_);
    if (value_elem_type != 1 && value_elem_type != 2) return FBBEV_E_BADARG;
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0)
        return FBBEV_E_BADARG;
    if (Za > FBBEV_DA_MAX_ZA || P % Za != 0) return FBBEV_E_UNSUPPORTED;
    if (dstep == 0.f) return FBBEV_E_BADARG;
    const long long units = (long long)B * Q * M;
    if (units == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth ||
        !offsets || !attn || !slots) return FBBEV_E_BADARG;
    const int HS = head_stride;
    // chunk-major rows of 8-element pieces covering Dh; 32-bit byte offsets; 8-byte aligned offsets / slots
    if (!(head_minor & 4) || HS % 8 != 0 || HS < (Dh + 7) / 8 * 8 || !aligned16(value) ||
        (((uintptr_t)offsets | (uintptr_t)slots) & 7) != 0 || (long long)B * Ncam * S * M * HS * 2 >= (1ll << 32) ||
        !(Dh == 10 || Dh == 8 || Dh == 16 || Dh == 32)) return FBBEV_E_UNSUPPORTED;
    long long ub = ((units + 255) / 256 + 7) / 8 * 8;
    if (ub > 65536) ub = 65536;
    const int LP = L * P;
    const bool stage = !(head_minor & 2) && LP % 4 == 0 && LP <= 36 && aligned16(attn);
    const size_t lds = stage ? (size_t)256 * (LP + 1) * sizeof(float) : 0;
#define FBBEV_DA_UNIT16(DH_, ET_)                                                                                    \
    FBBEV_LAUNCH((k_da_cross_attn_fwd_unit<DH_, true, true, ET_>), ub, 256, lds, (fbbev_rt_stream)stream_, units, value, \
                 spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, B, Ncam, S, M,  \
                 L, Q, P, Za, DC, d0, dstep, head_minor & 3, HS, stage ? 1 : 0, slots)
#define FBBEV_DA_UNIT16_ET(DH_) do { if (value_elem_type == 1) FBBEV_DA_UNIT16(DH_, 1); else FBBEV_DA_UNIT16(DH_, 2); } while (0)
    if (Dh == 10) FBBEV_DA_UNIT16_ET(10);
    else if (Dh == 8) FBBEV_DA_UNIT16_ET(8);
    else if (Dh == 16) FBBEV_DA_UNIT16_ET(16);
    else FBBEV_DA_UNIT16_ET(32);
#undef FBBEV_DA_UNIT16_ET
#undef FBBEV_DA_UNIT16
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_da_cross_attn_fwd(const float* value, const int64_t* spatial_shapes,
                                       const int64_t* level_start_index, const float* pred_depth,
                                       const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                       const float* offsets, const float* attn, int B, int Ncam, int S, int M,
                                       int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                       int head_minor, int head_stride, float* slots, fbbev_stream_t stream_) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0)
        return FBBEV_E_BADARG;
    if (Za > FBBEV_DA_MAX_ZA || P % Za != 0) return FBBEV_E_UNSUPPORTED;
    if (dstep == 0.f) return FBBEV_E_BADARG;
    const long long n = (long long)B * Q * M * Dh;
    if (n == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth ||
        !offsets || !attn || !slots) return FBBEV_E_BADARG;
    const int HS = head_stride == 0 ? Dh : head_stride;
    if (HS < Dh) return FBBEV_E_BADARG;
    const bool wide = HS % 4 == 0 && HS >= (Dh + 3) / 4 * 4 && aligned16(value);
    const bool qi = (head_minor & 4) != 0;                         // value rows stored [chunk][head][4 floats]
    if (qi && !wide) return FBBEV_E_UNSUPPORTED;
    const bool al8 = (((uintptr_t)value | (uintptr_t)offsets | (uintptr_t)slots) & 7) == 0;
    const bool fits32 = (long long)B * Ncam * S * M * HS * 4 < (1ll << 32);        // unit kernels: 32-bit byte offsets into value
    if (qi && !fits32) return FBBEV_E_UNSUPPORTED;
    if (al8 && fits32 && HS % 2 == 0 && (Dh == 10 || Dh == 8 || Dh == 16 || Dh == 32)) {   // a lane owns all Dh channels of a (b,q,head) unit
        const long long units = (long long)B * Q * M;
        long long ub = ((units + 255) / 256 + 7) / 8 * 8;         // XCD-contiguous order: a multiple of 8 workgroups
        if (ub > 65536) ub = 65536;
        // attention weights in (B,Q,M,L,P) staged through LDS ([256][L*P+1] floats) when 4 workgroups per CU still fit
        const int LP = L * P;
        const bool stage = !(head_minor & 2) && LP % 4 == 0 && LP <= 36 && aligned16(attn);
        const size_t lds = stage ? (size_t)256 * (LP + 1) * sizeof(float) : 0;
#define FBBEV_DA_UNIT_W(DH_, W_, Q_)                                                                                \
    FBBEV_LAUNCH((k_da_cross_attn_fwd_unit<DH_, W_, Q_>), ub, 256, lds, (fbbev_rt_stream)stream_, units, value,      \
                 spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, B, Ncam, S, M, \
                 L, Q, P, Za, DC, d0, dstep, head_minor & 3, HS, stage ? 1 : 0, slots)
#define FBBEV_DA_UNIT(DH_) do { if (qi) FBBEV_DA_UNIT_W(DH_, true, true); else if (wide) FBBEV_DA_UNIT_W(DH_, true, false); \
                                else FBBEV_DA_UNIT_W(DH_, false, false); } while (0)
        if (Dh == 10) FBBEV_DA_UNIT(10);
        else if (Dh == 8) FBBEV_DA_UNIT(8);
        else if (Dh == 16) FBBEV_DA_UNIT(16);
        else FBBEV_DA_UNIT(32);
#undef FBBEV_DA_UNIT_W
#undef FBBEV_DA_UNIT
        FBBEV_CHECK_LAUNCH();
        return 0;
    }
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FBBEV_LAUNCH(k_da_cross_attn_fwd, blocks, 256, 0, (fbbev_rt_stream)stream_, n, value, spatial_shapes,
                 level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, B, Ncam, S, M, Dh, L, Q, P,
                 Za, DC, d0, dstep, head_minor & 7, HS, slots);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

static bool da_pipe_disabled() {
    static const bool off = [] { const char* e = getenv("FBBEV_DA_PIPE"); return e && atoi(e) == 0; }();   // A/B timing knob, read once
    return off;
}
static bool da_pipe_shape_ok(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int head_minor, int HS) {
    const long long tokens = (long long)B * Ncam * S;
    const int LP = L * P;
    return (long long)B * Q * M > 0 && (head_minor & 7) == (1 | 4) &&                // head-minor offsets, (B,Q,M,L,P) attn, chunk-major rows
           Za == 4 && P % 4 == 0 && (Dh == 8 || Dh == 10) && HS == (Dh + 3) / 4 * 4 && LP % 4 == 0 && LP <= 36 &&
           (tokens + 1) * M * HS * 4 < (1ll << 32);                                  // 32-bit byte offsets incl. the zero token
}
extern "C" int fbbev_da_cross_attn_fwd_zt_fuses_softmax(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za,
                                                        int head_minor, int head_stride) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0 || Za <= 0) return 0;
    return (!da_pipe_disabled() && da_pipe_shape_ok(B, Ncam, S, M, Dh, L, Q, P, Za, head_minor, head_stride == 0 ? Dh : head_stride)) ? 1 : 0;
}

// Pipelined forward (k_da_cross_attn_fwd_pipe): the value buffer holds ONE MORE token than B*Ncam*S -- token index
// B*Ncam*S, M*HS floats, all +0.0f -- which padded corners and out-of-image samples read instead of branching around their
// loads.  Shapes outside the pipelined kernel's preconditions run fbbev_da_cross_attn_fwd on the same buffer.
extern "C" int fbbev_da_cross_attn_fwd_zt(const float* value, const int64_t* spatial_shapes,
                                          const int64_t* level_start_index, const float* pred_depth,
                                          const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                          const float* offsets, const float* attn, int B, int Ncam, int S, int M,
                                          int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                          int head_minor, int head_stride, int bev_w, float* slots,
                                          fbbev_stream_t stream_) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0 || bev_w < 0)
        return FBBEV_E_BADARG;
    const int HS = head_stride == 0 ? Dh : head_stride;
    const int LP = L * P;
    const long long units = (long long)B * Q * M;
    const long long tokens = (long long)B * Ncam * S;
    const bool pipe = value && offsets && attn && slots && dstep != 0.f &&
                      da_pipe_shape_ok(B, Ncam, S, M, Dh, L, Q, P, Za, head_minor, HS) && aligned16(attn) && aligned16(value) &&
                      (((uintptr_t)offsets | (uintptr_t)slots) & 7) == 0;
    if (!pipe || da_pipe_disabled()) {
        if (head_minor & FBBEV_DA_ATTN_LOGITS) return FBBEV_E_UNSUPPORTED;     // only the pipelined kernel fuses the softmax
        return fbbev_da_cross_attn_fwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                                       attn, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, head_stride, slots,
                                       stream_);
    }
    if (!spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth) return FBBEV_E_BADARG;
    // patch mapping (a workgroup = the 8 heads of an 8 x 4 patch of the BEV grid): needs the grid's row length
    static const bool patch_off = [] { const char* e = getenv("FBBEV_DA_PATCH"); return e && atoi(e) == 0; }();   // A/B timing knob, read once
    const int pw = (!patch_off && bev_w > 0 && M == 8 && Q % bev_w == 0) ? bev_w : 0;
    const long long wgs = pw ? (long long)B * ((pw + 7) / 8) * ((Q / pw + 3) / 4) : (units + 255) / 256;
    long long ub = (wgs + 7) / 8 * 8;                             // XCD-contiguous order: a multiple of 8 workgroups
    if (ub > 65536) ub = 65536;
    const size_t lds = (size_t)256 * (LP + 1) * sizeof(float);
    const unsigned zero_bytes = (unsigned)(tokens * M * HS * 4);
#define FBBEV_DA_PIPE(DH_, WPS_)                                                                                     \
    FBBEV_LAUNCH((k_da_cross_attn_fwd_pipe<DH_, 4, WPS_>), ub, 256, lds, (fbbev_rt_stream)stream_, units, value,         \
                 spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, B, Ncam, S, M, L, \
                 Q, P, DC, d0, dstep, HS, zero_bytes, (head_minor & FBBEV_DA_ATTN_LOGITS) ? 1 : 0, pw, slots)
    static const int wps = [] { const char* e = getenv("FBBEV_DA_PIPE_WPS"); return e ? atoi(e) : 2; }();   // tuning knob, read once (3: 168 registers, spills outside the loop)
    if (Dh == 10) { if (wps == 2) FBBEV_DA_PIPE(10, 2); else FBBEV_DA_PIPE(10, 3); }
    else { if (wps == 2) FBBEV_DA_PIPE(8, 2); else FBBEV_DA_PIPE(8, 3); }
#undef FBBEV_DA_PIPE
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ---- fbbev_da_cross_attn_fused: query rows -> slots in one kernel (da_fused_kernels.h)
static bool da_fused_shape_ok(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int bev_w) {
    if (M != 8 || (Dh != 10 && Dh != 8) || P != FBBEV_DAF_P || Za != FBBEV_DAF_ZA || L < 1 || bev_w <= 0 || Q % bev_w != 0 || Ncam > 32) return false;
    if (L > FBBEV_DAF_MAXL || fbbev_daf_lds_bytes(M * Dh, 8, Ncam) > 160 * 1024) return false;
    return (long long)S * Dh * 4 < (1ll << 31) && S < (1 << 24);                     // 32-bit byte offsets inside a head plane, 24-bit token indices
}
extern "C" int fbbev_da_cross_attn_fused_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int bev_w) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0 || Za <= 0) return 0;
    static const bool off = [] { const char* e = getenv("FBBEV_DA_FUSED"); return e && atoi(e) == 0; }();   // A/B timing knob, read once
    return (!off && da_fused_shape_ok(B, Ncam, S, M, Dh, L, Q, P, Za, bev_w)) ? 1 : 0;
}
static int da_cross_attn_fused_impl(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                    const float* query, long long query_row_stride, const float* addend,
                                    long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                                    const float* offsets_bias, const void* attn_fragments, const float* attn_bias, int B,
                                    int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                    int bev_w, int min_level_width, float* slots, fbbev_stream_t stream_, fbbev_daf_outproj op,
                                    int elem_type = 0) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0 || bev_w < 0 ||
        elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (elem_type != 0 && op.w_frag) return FBBEV_E_UNSUPPORTED;          // 16-bit planes: the plain sampler only
    if (Q == 0) return 0;
    if (!planes || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth || !query ||
        !offsets_fragments || !offsets_bias || !attn_fragments || !attn_bias || !slots || dstep == 0.f) return FBBEV_E_BADARG;
    const int E = M * Dh;
    if (query_row_stride == 0) query_row_stride = E;
    if (query_row_stride < E) return FBBEV_E_BADARG;
    if (addend) {
        if (addend_period <= 0) return FBBEV_E_BADARG;
        if (addend_row_stride == 0) addend_row_stride = E;
        if (addend_row_stride < E) return FBBEV_E_BADARG;
    } else { addend_row_stride = 0; addend_period = 1; }
    // min_level_width: the caller's host-side knowledge of the narrowest level (the x-corner runs are clamped into a row of at
    // least two tokens); the shapes themselves stay on the device
    if (!da_fused_shape_ok(B, Ncam, S, M, Dh, L, Q, P, Za, bev_w) || min_level_width < 2) return FBBEV_E_UNSUPPORTED;
    if (query_row_stride % 4 != 0 || addend_row_stride % 4 != 0 || !aligned16(query) || (addend && !aligned16(addend)) ||
        !aligned16(offsets_fragments) || !aligned16(attn_fragments) || !aligned16(offsets_bias) || ((uintptr_t)planes & 7) != 0 ||
        ((uintptr_t)slots & 7) != 0)
        return FBBEV_E_UNSUPPORTED;
    // heads per workgroup: 4 = two 256-thread workgroups per patch and per CU (one's prologue + projections under the other's
    // samples), 8 = one 512-thread workgroup per patch (FBBEV_DA_FUSED_HW, read once)
    auto read_hw = [] { const char* e = getenv("FBBEV_DA_FUSED_HW"); const int v = e ? atoi(e) : 4; return v == 8 ? 8 : 4; };
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: the tests run both forms inside one process
    const int hw_env = read_hw();
#else
    static const int hw_env = read_hw();
#endif
    const int hw = op.w_frag ? 8 : (elem_type ? 4 : hw_env); // the output_proj + LayerNorm tail needs all 8 heads of a query in one workgroup
    const long long wgs = (long long)B * ((bev_w + 7) / 8) * ((Q / bev_w + 7) / 8) * (8 / hw);
    const long long grid = (wgs + 7) / 8 * 8;
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    // round 5: floats of a wave's LDS staging region for coarse levels (the 8 x 22 level of BASELINE configs[2] at Dh = 10: 1 760);
    // a level that fits is sampled from LDS instead of through the vector L1.  FBBEV_DA_FUSED_STAGE=0 turns it off (A/B knob)
    auto read_stage = [] { const char* e = getenv("FBBEV_DA_FUSED_STAGE"); const int v = e ? atoi(e) : 1760; return v < 0 ? 0 : (v > 8192 ? 8192 : v); };
#ifdef FBBEV_TEST_OVERRIDES
    const int stage_floats = read_stage();
#else
    static const int stage_floats = read_stage();
#endif
    // (fbbev_da_cross_attn_fused_supported budgets LDS WITHOUT the staging region -- a launch whose staged layout does not
    // fit (8 heads per workgroup with ~25+ cameras) runs unstaged instead of failing after the probe said yes)
    int stage_eff = stage_floats;
    if (fbbev_daf_lds_bytes(E, hw, Ncam, stage_eff) > 160 * 1024) stage_eff = 0;
    const size_t lds = fbbev_daf_lds_bytes(E, hw, Ncam, stage_eff);
    if (lds > 160 * 1024) return FBBEV_E_UNSUPPORTED;
    static const bool pre_off = [] { const char* e = getenv("FBBEV_DA_FUSED_PRE"); return e && atoi(e) == 0; }();   // A/B knob, read once
    static const int diag = [] {                                                         // timing diagnostics (wrong results), read once
        const char* e = getenv("FBBEV_DA_FUSED_DIAG");
        const int v = e ? (atoi(e) & 63) : 0;
        if (v) fprintf(stderr, "libfbbev_hip: FBBEV_DA_FUSED_DIAG=%d -- fbbev_da_cross_attn_fused runs its timing-diagnostic build: RESULTS ARE WRONG BY DESIGN\n", v);
        return v;
    }();
    const int stage_arg = stage_eff | (pre_off ? 0x40000000 : 0) | (diag << 24);
#define FBBEV_DA_FUSED(DH_, NP_, HW_) FBBEV_DA_FUSED3(DH_, NP_, HW_, false, 0)
#define FBBEV_DA_FUSED2(DH_, NP_, HW_, OP_) FBBEV_DA_FUSED3(DH_, NP_, HW_, OP_, 0)
#define FBBEV_DA_FUSED3(DH_, NP_, HW_, OP_, ET_)                                                                       \
    do {                                                                                                               \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_da_cross_attn_fused<DH_, 8, NP_, HW_, OP_, ET_>, lds);          \
        if (e) return e;                                                                                               \
        FBBEV_LAUNCH((k_da_cross_attn_fused<DH_, 8, NP_, HW_, OP_, ET_>), grid, 64 * HW_, lds, (fbbev_rt_stream)stream_, planes, spatial_shapes, \
                     level_start_index, pred_depth, ref_cam, mask, qdepth, query, query_row_stride, addend,           \
                     addend_row_stride, addend_period, static_cast<const unsigned short*>(offsets_fragments),         \
                     offsets_bias, static_cast<const unsigned short*>(attn_fragments), attn_bias, B, Ncam, S, L, Q,    \
                     bev_w, DC, d0, dstep, slots, op, stage_arg);                                                      \
    } while (0)
    // (three samples in flight per lane -- FBBEV_DA_FUSED_NP=3 in round 4 -- measured no gain and, with the staged levels' second
    // sample loop, no longer fits the register file: the instantiations are gone, two in flight is the form)
    // (round 5, 16-bit planes: three / four samples in flight per lane -- 234 registers / 256 with 5 spills -- measured 401.3 / 405.1 us
    // against 402.1 us with two at BASELINE configs[2]: the sampler is not waiting for its gathers; profiles/r05_exp_da_fused_where.md)
    if (elem_type == 1) {              // round 5: camera tokens stored as bf16 / fp16 head planes (fp32 products and sums)
        if (Dh == 10) FBBEV_DA_FUSED3(10, 2, 4, false, 1); else FBBEV_DA_FUSED3(8, 2, 4, false, 1);
    } else if (elem_type == 2) {
        if (Dh == 10) FBBEV_DA_FUSED3(10, 2, 4, false, 2); else FBBEV_DA_FUSED3(8, 2, 4, false, 2);
    } else if (op.w_frag) {
        if (Dh == 10) FBBEV_DA_FUSED2(10, 2, 8, true); else FBBEV_DA_FUSED2(8, 2, 8, true);
    } else if (hw == 8) {
        if (Dh == 10) FBBEV_DA_FUSED(10, 2, 8); else FBBEV_DA_FUSED(8, 2, 8);
    } else if (diag && Dh == 10) {     // FBBEV_DA_FUSED_DIAG: the instantiation with the timing-diagnostic bits (wrong results by design)
        int e = fbbev_rt_allow_dyn_lds((const void*)k_da_cross_attn_fused<10, 8, 2, 4, false, 0, true>, lds);
        if (e) return e;
        FBBEV_LAUNCH((k_da_cross_attn_fused<10, 8, 2, 4, false, 0, true>), grid, 256, lds, (fbbev_rt_stream)stream_, planes, spatial_shapes,
                     level_start_index, pred_depth, ref_cam, mask, qdepth, query, query_row_stride, addend, addend_row_stride, addend_period,
                     static_cast<const unsigned short*>(offsets_fragments), offsets_bias, static_cast<const unsigned short*>(attn_fragments),
                     attn_bias, B, Ncam, S, L, Q, bev_w, DC, d0, dstep, slots, op, stage_arg);
    } else {
        if (Dh == 10) FBBEV_DA_FUSED(10, 2, 4); else FBBEV_DA_FUSED(8, 2, 4);
    }
#undef FBBEV_DA_FUSED
#undef FBBEV_DA_FUSED2
#undef FBBEV_DA_FUSED3
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_da_cross_attn_fused(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                         const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                         const float* query, long long query_row_stride, const float* addend,
                                         long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                                         const float* offsets_bias, const void* attn_fragments, const float* attn_bias, int B,
                                         int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                         int bev_w, int min_level_width, float* slots, fbbev_stream_t stream_) {
    return da_cross_attn_fused_impl(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, query,
                                    query_row_stride, addend, addend_row_stride, addend_period, offsets_fragments, offsets_bias,
                                    attn_fragments, attn_bias, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, bev_w, min_level_width,
                                    slots, stream_, fbbev_daf_outproj{nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0.f});
}
// The same kernel on 16-bit head planes (elem_type 1 bf16, 2 fp16; 0 = the entry above): the camera-token STORAGE option of the
// cross-attention (fbbev_rows_linear_x3_planes_e writes them); every product and sum stays fp32.
extern "C" int fbbev_da_cross_attn_fused_e(const void* planes, int elem_type, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                         const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                         const float* query, long long query_row_stride, const float* addend,
                                         long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                                         const float* offsets_bias, const void* attn_fragments, const float* attn_bias, int B,
                                         int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                                         int bev_w, int min_level_width, float* slots, fbbev_stream_t stream_) {
    return da_cross_attn_fused_impl(static_cast<const float*>(planes), spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth,
                                    query, query_row_stride, addend, addend_row_stride, addend_period, offsets_fragments, offsets_bias,
                                    attn_fragments, attn_bias, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, bev_w, min_level_width,
                                    slots, stream_, fbbev_daf_outproj{nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0.f}, elem_type);
}
// ... followed, inside the same workgroups (8 heads per workgroup), by output_proj + residual + LayerNorm: out rows instead of slots
extern "C" int fbbev_da_cross_attn_fused_ln(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                            const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                            const float* query, long long query_row_stride, const float* addend,
                                            long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                                            const float* offsets_bias, const void* attn_fragments, const float* attn_bias,
                                            const void* out_fragments, const float* out_bias, const float* residual,
                                            long long residual_row_stride, const float* ln_weight, const float* ln_bias, float ln_eps,
                                            int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0,
                                            float dstep, int bev_w, int min_level_width, float* out, fbbev_stream_t stream_) {
    if (!out_fragments || !out_bias || !ln_weight || !ln_bias || !(ln_eps >= 0.f) || M <= 0 || Dh <= 0) return FBBEV_E_BADARG;
    const int E = M * Dh;
    if (residual && residual_row_stride == 0) residual_row_stride = E;
    if (residual && residual_row_stride < E) return FBBEV_E_BADARG;
    if (E % 16 != 0 || !aligned16(out_fragments) || !aligned16(out_bias) || !aligned16(ln_weight) || !aligned16(ln_bias) ||
        !aligned16(out) || (residual && (!aligned16(residual) || residual_row_stride % 4 != 0))) return FBBEV_E_UNSUPPORTED;
    return da_cross_attn_fused_impl(planes, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, query,
                                    query_row_stride, addend, addend_row_stride, addend_period, offsets_fragments, offsets_bias,
                                    attn_fragments, attn_bias, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, bev_w, min_level_width,
                                    out, stream_,
                                    fbbev_daf_outproj{static_cast<const unsigned short*>(out_fragments), out_bias, residual,
                                                      residual_row_stride, ln_weight, ln_bias, ln_eps});
}

// ---- Z-mean / Z-sum of a materialised (B*C, Z, Y*X) volume (training path; k_volume_zreduce)
extern "C" int fbbev_volume_zreduce(const float* volume, long long n_bc, int Z, long long YX, float divisor, float* out,
                                    fbbev_stream_t stream_) {
    if (n_bc < 0 || Z <= 0 || YX <= 0 || divisor == 0.f) return FBBEV_E_BADARG;
    if (n_bc == 0) return 0;
    if (!volume || !out) return FBBEV_E_BADARG;
    if (YX % 4 != 0 || !aligned16(volume) || !aligned16(out)) return FBBEV_E_UNSUPPORTED;
    const long long items = n_bc * (YX / 4);
    if ((items + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_volume_zreduce, (items + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, volume, n_bc, Z, YX, divisor, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_volume_zreduce_inner(const float* volume, long long n_pillars, int Z, float divisor, float* out,
                                          fbbev_stream_t stream_) {
    if (n_pillars < 0 || Z <= 0 || divisor == 0.f) return FBBEV_E_BADARG;
    if (n_pillars == 0) return 0;
    if (!volume || !out) return FBBEV_E_BADARG;
    if (Z % 4 != 0 || !aligned16(volume) || (n_pillars + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_volume_zreduce_inner, (n_pillars + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, volume, n_pillars, Z, divisor, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}
extern "C" int fbbev_volume_z_to_front(const float* src, long long n_bc, int Z, long long YX, float* dst, fbbev_stream_t stream_) {
    if (n_bc < 0 || Z <= 0 || YX <= 0) return FBBEV_E_BADARG;
    if (n_bc == 0) return 0;
    if (!src || !dst || src == dst) return FBBEV_E_BADARG;
    if (Z % 4 != 0 || !aligned16(src) || (n_bc * YX + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_volume_z_to_front, (n_bc * YX + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, src, n_bc, Z, YX, dst);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ---- training forward on head planes (da_bwd_planes_kernels.h): offsets / softmaxed weights from memory, tokens as planes
extern "C" int fbbev_value_rows_to_head_planes(const float* value, long long n_tokens, int S, int M, int Dh, int head_stride,
                                               int interleaved, float* planes, fbbev_stream_t stream_) {
    if (n_tokens < 0 || S <= 0 || M <= 0 || Dh <= 0 || n_tokens % S != 0) return FBBEV_E_BADARG;
    if (n_tokens == 0) return 0;
    const int HS = head_stride == 0 ? Dh : head_stride;
    if (!value || !planes || HS < Dh || (interleaved && HS % 4 != 0)) return FBBEV_E_BADARG;
    const long long n_el = n_tokens * M * Dh;
    if ((n_el + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_value_rows_to_head_planes, (n_el + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, value, n_tokens, S, M, Dh, HS,
                 interleaved ? 1 : 0, planes);
    FBBEV_CHECK_LAUNCH();
    return 0;
}
extern "C" int fbbev_da_cross_attn_fwd_planes_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0 || Za <= 0) return 0;
    static const bool off = [] { const char* e = getenv("FBBEV_DA_FWD_PLANES"); return e && atoi(e) == 0; }();   // A/B timing knob, read once
    return (!off && M == 8 && (Dh == 10 || Dh == 8) && P == FBBEV_DAF_P && Za == FBBEV_DAF_ZA && (long long)S * Dh * 4 < (1ll << 31) &&
            fbbev_dfp_lds_bytes(Ncam) <= 156 * 1024) ? 1 : 0;
}
extern "C" int fbbev_da_cross_attn_fwd_planes(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                              const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                              const float* offsets, const float* attn, int B, int Ncam, int S, int M, int Dh, int L,
                                              int Q, int P, int Za, int DC, float d0, float dstep, int head_minor, int bev_w,
                                              int min_level_width, float* slots, fbbev_stream_t stream_) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0 || bev_w < 0)
        return FBBEV_E_BADARG;
    if (Q == 0) return 0;
    if (!planes || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth || !offsets || !attn ||
        !slots || dstep == 0.f) return FBBEV_E_BADARG;
    if (!fbbev_da_cross_attn_fwd_planes_supported(B, Ncam, S, M, Dh, L, Q, P, Za) || min_level_width < 2 ||
        ((uintptr_t)planes & 7) != 0 || ((uintptr_t)slots & 7) != 0 || ((uintptr_t)offsets & 7) != 0 || !aligned16(ref_cam) ||
        !aligned16(qdepth) || ((uintptr_t)mask & 3) != 0) return FBBEV_E_UNSUPPORTED;
    const int gw = (bev_w > 0 && Q % bev_w == 0) ? bev_w : 0;
    const long long wgs = gw > 0 ? (long long)B * ((gw + 7) / 8) * ((Q / gw + 7) / 8) : (long long)B * ((Q + 63) / 64);
    const long long grid = (wgs + 7) / 8 * 8;
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = fbbev_dfp_lds_bytes(Ncam);
#define FBBEV_DA_FWD_PLANES(DH_)                                                                                       \
    do {                                                                                                               \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_da_fwd_planes<DH_, 8>, lds);                                     \
        if (e) return e;                                                                                               \
        FBBEV_LAUNCH((k_da_fwd_planes<DH_, 8>), grid, 512, lds, (fbbev_rt_stream)stream_, planes, spatial_shapes,       \
                     level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, B, Ncam, S, L, Q, gw, DC, d0,  \
                     dstep, head_minor & 3, slots);                                                                    \
    } while (0)
    if (Dh == 10) FBBEV_DA_FWD_PLANES(10); else FBBEV_DA_FWD_PLANES(8);
#undef FBBEV_DA_FWD_PLANES
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ---- fbbev_msda_self_fused: BEV self-attention, query rows -> attention output in one kernel (da_fused_kernels.h)
extern "C" int fbbev_msda_self_fused_supported(int B, int S, int M, int Dh, int L, int Q, int P, int bev_w) {
    if (B <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0) return 0;
    static const bool off = [] { const char* e = getenv("FBBEV_MSDA_FUSED"); return e && atoi(e) == 0; }();   // A/B timing knob, read once
    return (!off && M == 8 && (Dh == 10 || Dh == 8) && L == 1 && P == FBBEV_MSF_P && bev_w > 0 && Q % bev_w == 0 &&
            (long long)S * Dh * 4 < (1ll << 31)) ? 1 : 0;
}
static int msda_self_fused_impl(const float* planes, const float* reference_points, const float* query,
                                long long query_row_stride, const float* addend, long long addend_row_stride,
                                long long addend_period, const void* offsets_fragments, const float* offsets_bias,
                                const void* attn_fragments, const float* attn_bias, int B, int S, int M, int Dh, int L, int Q,
                                int P, int bev_w, int level_h, int level_w, float* out, fbbev_stream_t stream_,
                                fbbev_daf_outproj op) {
    if (B <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || bev_w < 0 || level_h <= 0 || level_w <= 0)
        return FBBEV_E_BADARG;
    if (Q == 0) return 0;
    if (!planes || !reference_points || !query || !offsets_fragments || !offsets_bias || !attn_fragments || !attn_bias || !out)
        return FBBEV_E_BADARG;
    if ((long long)level_h * level_w != S) return FBBEV_E_BADARG;
    const int E = M * Dh;
    if (query_row_stride == 0) query_row_stride = E;
    if (query_row_stride < E) return FBBEV_E_BADARG;
    if (addend) {
        if (addend_period <= 0) return FBBEV_E_BADARG;
        if (addend_row_stride == 0) addend_row_stride = E;
        if (addend_row_stride < E) return FBBEV_E_BADARG;
    } else { addend_row_stride = 0; addend_period = 1; }
    if (!(M == 8 && (Dh == 10 || Dh == 8) && L == 1 && P == FBBEV_MSF_P && bev_w > 0 && Q % bev_w == 0) || level_w < 2 ||
        (long long)S * Dh * 4 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (query_row_stride % 4 != 0 || addend_row_stride % 4 != 0 || !aligned16(query) || (addend && !aligned16(addend)) ||
        !aligned16(offsets_fragments) || !aligned16(attn_fragments) || ((uintptr_t)planes & 7) != 0 || ((uintptr_t)out & 7) != 0 ||
        ((uintptr_t)reference_points & 7) != 0) return FBBEV_E_UNSUPPORTED;
    const long long wgs = (long long)B * ((bev_w + 7) / 8) * ((Q / bev_w + 7) / 8);
    const long long grid = (wgs + 7) / 8 * 8;
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = fbbev_msf_lds_bytes(E, M);
#define FBBEV_MSDA_SELF(DH_) do { if (op.w_frag) FBBEV_MSDA_SELF2(DH_, true); else FBBEV_MSDA_SELF2(DH_, false); } while (0)
#define FBBEV_MSDA_SELF2(DH_, OP_)                                                                                      \
    do {                                                                                                               \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_msda_self_fused<DH_, 8, OP_>, lds);                             \
        if (e) return e;                                                                                               \
        FBBEV_LAUNCH((k_msda_self_fused<DH_, 8, OP_>), grid, 512, lds, (fbbev_rt_stream)stream_, planes, reference_points,  \
                     query, query_row_stride, addend, addend_row_stride, addend_period,                                \
                     static_cast<const unsigned short*>(offsets_fragments), offsets_bias,                              \
                     static_cast<const unsigned short*>(attn_fragments), attn_bias, B, Q, bev_w, S, level_h, level_w, out, op); \
    } while (0)
    if (Dh == 10) FBBEV_MSDA_SELF(10); else FBBEV_MSDA_SELF(8);
#undef FBBEV_MSDA_SELF
#undef FBBEV_MSDA_SELF2
    FBBEV_CHECK_LAUNCH();
    return 0;
}
extern "C" int fbbev_msda_self_fused(const float* planes, const float* reference_points, const float* query,
                                     long long query_row_stride, const float* addend, long long addend_row_stride,
                                     long long addend_period, const void* offsets_fragments, const float* offsets_bias,
                                     const void* attn_fragments, const float* attn_bias, int B, int S, int M, int Dh, int L, int Q,
                                     int P, int bev_w, int level_h, int level_w, float* out, fbbev_stream_t stream_) {
    return msda_self_fused_impl(planes, reference_points, query, query_row_stride, addend, addend_row_stride, addend_period,
                                offsets_fragments, offsets_bias, attn_fragments, attn_bias, B, S, M, Dh, L, Q, P, bev_w, level_h,
                                level_w, out, stream_, fbbev_daf_outproj{nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0.f});
}
// ... followed, inside the same workgroups, by output_proj + residual + LayerNorm: out = LN(W_o attention + b_o + residual)
extern "C" int fbbev_msda_self_fused_ln(const float* planes, const float* reference_points, const float* query,
                                        long long query_row_stride, const float* addend, long long addend_row_stride,
                                        long long addend_period, const void* offsets_fragments, const float* offsets_bias,
                                        const void* attn_fragments, const float* attn_bias, const void* out_fragments,
                                        const float* out_bias, const float* residual, long long residual_row_stride,
                                        const float* ln_weight, const float* ln_bias, float ln_eps, int B, int S, int M, int Dh,
                                        int L, int Q, int P, int bev_w, int level_h, int level_w, float* out,
                                        fbbev_stream_t stream_) {
    if (!out_fragments || !out_bias || !ln_weight || !ln_bias || !(ln_eps >= 0.f) || M <= 0 || Dh <= 0) return FBBEV_E_BADARG;
    const int E = M * Dh;
    if (residual && residual_row_stride == 0) residual_row_stride = E;
    if (residual && residual_row_stride < E) return FBBEV_E_BADARG;
    if (E % 16 != 0 || !aligned16(out_fragments) || !aligned16(out_bias) || !aligned16(ln_weight) || !aligned16(ln_bias) ||
        !aligned16(out) || (residual && (!aligned16(residual) || residual_row_stride % 4 != 0))) return FBBEV_E_UNSUPPORTED;
    return msda_self_fused_impl(planes, reference_points, query, query_row_stride, addend, addend_row_stride, addend_period,
                                offsets_fragments, offsets_bias, attn_fragments, attn_bias, B, S, M, Dh, L, Q, P, bev_w, level_h,
                                level_w, out, stream_,
                                fbbev_daf_outproj{static_cast<const unsigned short*>(out_fragments), out_bias, residual,
                                                  residual_row_stride, ln_weight, ln_bias, ln_eps});
}

extern "C" int fbbev_rows_to_head_planes(const float* rows, long long n_rows, int tokens_per_image, int M, int Dh, float* planes,
                                         fbbev_stream_t stream_) {
    if (n_rows < 0 || tokens_per_image <= 0 || M <= 0 || Dh <= 0) return FBBEV_E_BADARG;
    if (n_rows == 0) return 0;
    if (!rows || !planes || n_rows % tokens_per_image != 0) return FBBEV_E_BADARG;
    const long long n = n_rows * M * Dh;
    if ((n + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_rows_to_head_planes, (n + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, rows, n_rows, tokens_per_image, M, Dh, planes);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_da_cross_attn_bwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                                       const float* qdepth, const float* offsets, const float* attn,
                                       const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                                       int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                                       float* grad_value, float* grad_pred_depth, float* grad_offsets,
                                       float* grad_attn, fbbev_stream_t stream_) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0)
        return FBBEV_E_BADARG;
    if (Za > FBBEV_DA_MAX_ZA || P % Za != 0) return FBBEV_E_UNSUPPORTED;
    if (dstep == 0.f) return FBBEV_E_BADARG;
    const long long units = (long long)B * Q * M;
    if (units == 0) return 0;
    if (!value || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth || !offsets ||
        !attn || !grad_slots || !grad_value || !grad_pred_depth || !grad_offsets || !grad_attn) return FBBEV_E_BADARG;
    if (Dh > 32) return FBBEV_E_UNSUPPORTED;
    const int HS = head_stride == 0 ? Dh : head_stride;
    if (HS < Dh) return FBBEV_E_BADARG;
    if ((head_minor & 4) && HS % 4 != 0) return FBBEV_E_UNSUPPORTED;
#define FBBEV_DA_BWD(GW_)                                                                                             \
    FBBEV_LAUNCH(k_da_cross_attn_bwd<GW_>, (units * GW_ + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, units, value,  \
                 spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots, B,  \
                 Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor & 7, HS, grad_value, grad_pred_depth,        \
                 grad_offsets, grad_attn)
    if ((units * 32 + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (Dh <= 16) FBBEV_DA_BWD(16);
    else FBBEV_DA_BWD(32);
#undef FBBEV_DA_BWD
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// LDS-plane backward (needs a caller-owned partial buffer): k_da_cross_attn_bwd_unit (unit-owned gradients) +
// k_da_cross_attn_bwd_scatter per token region + k_da_bwd_reduce
struct da_region { int lvl0, lvl1, tok0, tok1; };
struct da_bwd_plan { int chunks, q_per_chunk, threads, n_regions, info_stride; size_t lds, ws, ws_part; da_region reg[32]; };

// token regions of at most `budget` tokens: whole consecutive levels, or bands of rows of a level larger than the budget
// (level_hw: HOST array of L (h, w) pairs; without it only the one-region case -- the whole pyramid fits -- is planned)
static int da_bwd_regions(int L, const int32_t* level_hw, int S, int budget, da_region* out, int cap) {
    if (!level_hw) {
        if (S > budget) return 0;
        out[0] = {0, L, 0, S};
        return 1;
    }
    int n = 0, start = 0;
    da_region cur = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
        const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
        if (h <= 0 || w <= 0) return 0;
        const int cnt = h * w;
        if (cnt > budget) {
            if (cur.lvl1 > cur.lvl0) { if (n == cap) return 0; out[n++] = cur; }
            const int rows = budget / w;
            if (rows < 1) return 0;
            for (int r = 0; r < h; r += rows) {
                if (n == cap) return 0;
                out[n++] = {l, l + 1, start + r * w, start + (r + rows < h ? r + rows : h) * w};
            }
            cur = {l + 1, l + 1, start + cnt, start + cnt};
        } else if (cur.lvl1 > cur.lvl0 && (cur.tok1 - cur.tok0) + cnt > budget) {
            if (n == cap) return 0;
            out[n++] = cur;
            cur = {l, l + 1, start, start + cnt};
        } else {
            if (cur.lvl1 == cur.lvl0) cur = {l, l + 1, start, start + cnt};
            else { cur.lvl1 = l + 1; cur.tok1 = start + cnt; }
        }
        start += cnt;
    }
    if (cur.lvl1 > cur.lvl0) { if (n == cap) return 0; out[n++] = cur; }
    return start == S ? n : 0;
}

// Tuning / test overrides of the plan, read ONCE per process (function-local static: thread-safe) -- the Python forward
// derives the value-row layout from this planner and the backward launches from it, so a variable that changed between
// the two calls must not be able to make them disagree.  Unset = the measured defaults.
struct da_bwd_overrides { int tokens, chunks, threads, copies, lds_kb, prepass; };
static da_bwd_overrides da_bwd_read_env() {
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
    return da_bwd_overrides{num("FBBEV_DA_BWD_TOKENS"), num("FBBEV_DA_BWD_CHUNKS"), num("FBBEV_DA_BWD_THREADS"),
                            num("FBBEV_DA_BWD_COPIES"), num("FBBEV_DA_BWD_LDS_KB"),
                            getenv("FBBEV_DA_BWD_PREPASS") ? num("FBBEV_DA_BWD_PREPASS") : -1};
}
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build only (tests/emu/rt.h): the tests switch plans inside one process
static da_bwd_overrides da_bwd_env() { return da_bwd_read_env(); }
#else
static const da_bwd_overrides& da_bwd_env() {
    static const da_bwd_overrides o = da_bwd_read_env();
    return o;
}
#endif

// LDS bytes of the fixed-point planes of one workgroup: 68 KB = two workgroups per CU (FBBEV_DA_BWD_LDS_KB tunes it)
static int da_bwd_plane_kb() { const int v = da_bwd_env().lds_kb; return v >= 8 && v <= 144 ? v : 68; }

static bool da_bwd_tile_plan(int B, int Ncam, int S, int M, int Dh, int Q, int HS, int L, int P, const int32_t* level_hw,
                             da_bwd_plan* pl) {
    if (Dh > 16 || HS % 4 != 0 || HS > 16 || Q <= 0) return false;
    if (P > FBBEV_DA_BWD_MAXP || !(Dh == 10 || Dh == 8 || Dh == 16 || Dh == 4)) return false;
    int budget = (da_bwd_plane_kb() * 1024) / (HS * (int)sizeof(long long)) - 8;   // tokens per LDS plane (64-bit words, skewed)
    { const int v = da_bwd_env().tokens; if (v > 0 && v < budget) budget = v; }   // tests: force bands
    pl->n_regions = da_bwd_regions(L, level_hw, S, budget, pl->reg, 32);
    if (pl->n_regions == 0) return false;
    int max_tok = 0;
    for (int r = 0; r < pl->n_regions; ++r) max_tok = pl->reg[r].tok1 - pl->reg[r].tok0 > max_tok ? pl->reg[r].tok1 - pl->reg[r].tok0 : max_tok;
    const size_t plane = (size_t)FBBEV_DA_PLANE_WORDS(max_tok, HS) * sizeof(long long);
    // two workgroups per CU in one round (512 of them)
    long long want = (512 + (long long)B * M - 1) / ((long long)B * M);
    { const int v = da_bwd_env().chunks; if (v > 0) want = v; }
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    int qpc = (int)((Q + want - 1) / want);
    // one lane per query the camera sees: 512 threads (measured 0.177 vs 0.208 ms at the shipped shape) unless the chunk is short
    pl->threads = qpc >= 512 ? 512 : 256;
    { const int v = da_bwd_env().threads; if (v == 256 || v == 512) pl->threads = v; }
    const int ng = pl->threads;                                                   // queries per workgroup iteration
    qpc = (qpc + ng - 1) / ng * ng;
    if (qpc > 65535) return false;                                                // chunk-relative 16-bit query ids
    pl->q_per_chunk = qpc;
    pl->chunks = (Q + qpc - 1) / qpc;
    // + the camera's hit list, its counter, the block maximum
    pl->lds = plane + (size_t)((qpc + 1) & ~1) * 2 + (size_t)(1 + pl->threads / 64) * sizeof(int);
    if (pl->lds > (size_t)(da_bwd_plane_kb() + 12) * 1024) return false;
    pl->ws_part = ((size_t)B * M * pl->chunks * Ncam * S * HS * sizeof(float) + 255) / 256 * 256;
    // pre-pass (k_da_bwd_hitinfo; FBBEV_DA_BWD_PREPASS=0 turns it off): per (b, camera, query) the camera count and the Za depth
    // weights, behind the partials -- worth it when several region launches would each recompute them (configs[2] pyramid:
    // scatter 2.43 -> 2.22 ms, profiles/r03_exp_da_bwd_prepass.jsonl)
    pl->info_stride = 0;
    pl->ws = pl->ws_part;
    if (da_bwd_env().prepass != 0 && pl->n_regions > 1) {
        pl->info_stride = 8;                                        // 1 + Za <= 8 floats (Za is checked at the launch)
        pl->ws += (size_t)B * Ncam * Q * pl->info_stride * sizeof(float);
    }
    return true;
}

// ---- output-owned planes (k_da_bwd_scatter_owned; FBBEV_DA_BWD_OWNED=0 restores the chunked scatter above)
struct da_own_plan { fbbev_da_bwd_region_tab tab; int threads, info_stride, unit_planes; size_t lds, off_list, off_count, off_gmax, off_planes, ws; };
// FBBEV_DA_BWD_OWNED: 1 = whenever the shape is supported, 0 = never, unset = when the launch has at least one workgroup per CU
// (the shipped single-level shape at B = 4 has 192 (sample, camera, head) planes: the chunked scatter's 512 workgroups win there)
static int da_bwd_owned_mode() {
#ifdef FBBEV_TEST_OVERRIDES
    const char* e = getenv("FBBEV_DA_BWD_OWNED"); return e ? (atoi(e) != 0 ? 1 : 0) : -1;
#else
    static const int mode = [] { const char* e = getenv("FBBEV_DA_BWD_OWNED"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    return mode;
#endif
}
// LDS bytes of an owned plane: 136 KB = one 512-thread workgroup per CU; fewer, larger regions beat two workgroups per CU
// (configs[2] pyramid: 1.06 ms at 136 KB / 512 threads, 1.38 ms at 68 KB / 256, 2.18 ms at 68 KB / 512, 3.47 ms at 34 KB)
static int da_own_plane_kb() { const int v = da_bwd_env().lds_kb; return v >= 8 && v <= 144 ? v : 136; }
static bool da_own_plan_make(int B, int Ncam, int S, int M, int Dh, int Q, int HS, int L, int P, int Za, const int32_t* level_hw,
                             da_own_plan* pl) {
    const int mode = da_bwd_owned_mode();
    if (mode == 0) return false;
    if (Dh > 16 || HS % 4 != 0 || HS > 16 || Q <= 0 || P > FBBEV_DA_BWD_MAXP || !(Dh == 10 || Dh == 8 || Dh == 16 || Dh == 4)) return false;
    pl->info_stride = 0;
    if (Za > FBBEV_DA_HIT_ZA) return false;                             // a hit record holds 4 anchors
    const size_t budget_bytes = (size_t)da_own_plane_kb() * 1024;
    int budget = (int)(budget_bytes / (HS * sizeof(long long))) - 8;             // tokens per LDS plane (64-bit words, skewed)
    { const int v = da_bwd_env().tokens; if (v > 0 && v < budget) budget = v; }   // tests: force bands
    // one region per level (a level larger than the budget: bands of rows), so that every level gets the copies its size allows
    da_region reg[24];
    int n = 0;
    if (!level_hw) {
        if (S > budget) return false;
        reg[n++] = {0, L, 0, S};
    } else {
        int start = 0;
        for (int l = 0; l < L; ++l) {
            const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
            if (h <= 0 || w <= 0) return false;
            const int cnt = h * w, rows = cnt > budget ? budget / w : h;
            if (rows < 1) return false;
            for (int r = 0; r < h; r += rows) {
                if (n == 24) return false;
                reg[n++] = {l, l + 1, start + r * w, start + (r + rows < h ? r + rows : h) * w};
            }
            start += cnt;
        }
        if (start != S) return false;
    }
    // small regions first in the launch order: their adds collide most, they run longest
    // (measured: 1 121-1 128 us small-first against 1 152-1 154 us large-first, tools/sessions_r04/gpu_r04_s39_region_order.sh)
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && reg[j].tok1 - reg[j].tok0 < reg[j - 1].tok1 - reg[j - 1].tok0; --j) { const da_region t = reg[j]; reg[j] = reg[j - 1]; reg[j - 1] = t; }
    size_t lds = 0;
    pl->tab.n = n;
    pl->tab.perm = 0u;
    for (int i = 0; i < n; ++i) {
        // lane -> hit permutation (spreads neighbouring queries over the waves) for whole levels only: in a BAND of rows it costs
        // more than it saves -- consecutive hits sample neighbouring rows, so whole waves skip the corners outside the band (-5 %)
        const bool band = level_hw && (reg[i].tok1 - reg[i].tok0) != level_hw[2 * reg[i].lvl0] * level_hw[2 * reg[i].lvl0 + 1];
        if (!band) pl->tab.perm |= 1u << i;
        const size_t one = (size_t)FBBEV_DA_PLANE_WORDS(reg[i].tok1 - reg[i].tok0, HS) * sizeof(long long);
        int copies = (int)(budget_bytes / one);
        copies = copies < 1 ? 1 : (copies > 16 ? 16 : copies);
        { const int v = da_bwd_env().copies; if (v >= 1 && v < copies) copies = v; }
        pl->tab.lvl0[i] = reg[i].lvl0; pl->tab.lvl1[i] = reg[i].lvl1; pl->tab.tok0[i] = reg[i].tok0; pl->tab.tok1[i] = reg[i].tok1;
        pl->tab.copies[i] = copies;
        lds = one * copies > lds ? one * copies : lds;
    }
    if (lds > budget_bytes + 12 * 1024) return false;
    pl->lds = lds;
    pl->threads = 512;
    { const int v = da_bwd_env().threads; if (v == 256 || v == 512) pl->threads = v; }
    if ((long long)n * B * Ncam * M >= (1ll << 31) || (long long)B * Ncam * Q >= (1ll << 31)) return false;
    if (mode < 0 && (long long)n * B * Ncam * M < 256) return false;              // too few planes to fill the chip: chunked scatter
    pl->off_list = 0;                                                  // hit records [B*Ncam][Q][16 floats] at the start of ws
    pl->off_count = align_up((size_t)B * Ncam * Q * FBBEV_DA_HIT_REC * sizeof(float), 256);
    pl->off_gmax = pl->off_count + align_up((size_t)B * Ncam * sizeof(int), 256);
    pl->ws = pl->off_gmax + align_up((size_t)B * sizeof(unsigned int), 256);          // one fixed-point scale per sample
    // unit gradients on head planes (k_da_bwd_unit_planes, da_bwd_planes_kernels.h; FBBEV_DA_BWD_UNIT_PLANES=0 keeps the row kernel):
    // M = 8, Dh in {8, 10}, 8 points, 4 anchors, every level at least 2 tokens wide; the planes sit behind the hit lists
    pl->unit_planes = 0;
    pl->off_planes = pl->ws;
    {
#ifdef FBBEV_TEST_OVERRIDES
        const char* e = getenv("FBBEV_DA_BWD_UNIT_PLANES"); const bool on = !(e && atoi(e) == 0);
#else
        static const bool on = [] { const char* e = getenv("FBBEV_DA_BWD_UNIT_PLANES"); return !(e && atoi(e) == 0); }();
#endif
        bool wide = level_hw != nullptr;
        for (int l = 0; wide && l < L; ++l) wide = level_hw[2 * l + 1] >= 2;
        if (on && wide && M == 8 && (Dh == 10 || Dh == 8) && P == FBBEV_DAF_P && (long long)S * Dh * 4 < (1ll << 31) &&
            fbbev_dbp_lds_bytes(M, Ncam) <= 156 * 1024) {
            pl->unit_planes = 1;
            pl->ws += align_up((size_t)B * Ncam * M * S * Dh * sizeof(float), 256);
        }
    }
    return true;
}

extern "C" size_t fbbev_da_cross_attn_bwd_ws_bytes(int B, int Ncam, int S, int M, int Dh, int Q, int head_stride,
                                                   int num_levels, int num_points, const int32_t* level_hw_host) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || num_levels <= 0 || num_points <= 0) return 0;
    da_bwd_plan pl;
    const int HS = head_stride == 0 ? Dh : head_stride;
    if (HS < Dh) return 0;
    // Za is not an argument here (the owned plan needs Za <= 4 and word-aligned records, which only the launch can check): the
    // size covers BOTH routes, so a launch whose owned plan fails still finds room for the chunked LDS planes instead of falling
    // back to global atomics
    da_own_plan op;
    const size_t own = da_own_plan_make(B, Ncam, S, M, Dh, Q, HS, num_levels, num_points, 1, level_hw_host, &op) ? op.ws : 0;
    const size_t tile = da_bwd_tile_plan(B, Ncam, S, M, Dh, Q, HS, num_levels, num_points, level_hw_host, &pl) ? pl.ws : 0;
    return own > tile ? own : tile;
}

// The same query with the number of Z anchors the launch will see: exactly the size of the route that launch takes (owned planes
// when their plan holds for this Za, the chunked planes otherwise) instead of the maximum over both.
extern "C" size_t fbbev_da_cross_attn_bwd_ws_bytes_za(int B, int Ncam, int S, int M, int Dh, int Q, int head_stride, int num_levels,
                                                      int num_points, int num_z_anchors, const int32_t* level_hw_host) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || num_levels <= 0 || num_points <= 0 || num_z_anchors <= 0) return 0;
    const int HS = head_stride == 0 ? Dh : head_stride;
    if (HS < Dh) return 0;
    da_own_plan op;
    if (num_points % num_z_anchors == 0 &&
        da_own_plan_make(B, Ncam, S, M, Dh, Q, HS, num_levels, num_points, num_z_anchors, level_hw_host, &op)) return op.ws;
    da_bwd_plan pl;
    return da_bwd_tile_plan(B, Ncam, S, M, Dh, Q, HS, num_levels, num_points, level_hw_host, &pl) ? pl.ws : 0;
}

// (A) of both LDS-plane backward routes: unit-owned gradients (offsets / attention / depth distribution), the forward's launch shape
static int da_bwd_unit_launch(fbbev_rt_stream stream, const float* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                              const float* qdepth, const float* offsets, const float* attn, const float* grad_slots, int B,
                              int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                              int head_minor, int HS, float* grad_pred_depth, float* grad_offsets, float* grad_attn,
                              unsigned int* gmax_bits) {
    // (A) unit-owned gradients, the forward's launch shape
    const long long units = (long long)B * Q * M;
    long long ub = ((units + 255) / 256 + 7) / 8 * 8;
    if (ub > 65536) ub = 65536;
    const size_t lds_a = (size_t)256 * (4 * P + 1) * sizeof(float);
    const bool qi = (head_minor & 4) != 0;
#define FBBEV_DA_BWD_UNIT(DH_)                                                                                          \
    do {                                                                                                                \
    if (qi) FBBEV_LAUNCH((k_da_cross_attn_bwd_unit<DH_, true>), ub, 256, lds_a, stream, units, value, spatial_shapes, \
                         level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots, B, Ncam, S, M, \
                         L, Q, P, Za, DC, d0, dstep, head_minor & 3, HS, grad_pred_depth, grad_offsets, grad_attn, gmax_bits); \
    else FBBEV_LAUNCH((k_da_cross_attn_bwd_unit<DH_, false>), ub, 256, lds_a, stream, units, value, spatial_shapes, \
                      level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots, B, Ncam, S, M, \
                      L, Q, P, Za, DC, d0, dstep, head_minor & 3, HS, grad_pred_depth, grad_offsets, grad_attn, gmax_bits); \
    } while (0)
    if (Dh == 10) FBBEV_DA_BWD_UNIT(10);
    else if (Dh == 8) FBBEV_DA_BWD_UNIT(8);
    else if (Dh == 4) FBBEV_DA_BWD_UNIT(4);
    else FBBEV_DA_BWD_UNIT(16);
#undef FBBEV_DA_BWD_UNIT
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// output-owned route: unit gradients, then init + hit lists + ONE scatter launch over every token region
static int da_bwd_owned_launch(const da_own_plan& op, fbbev_rt_stream stream, const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                               const float* qdepth, const float* offsets, const float* attn, const float* grad_slots, int B,
                               int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                               int head_minor, int HS, float* grad_value, float* grad_pred_depth, float* grad_offsets,
                               float* grad_attn, void* ws, int bev_w, const float* planes_in = nullptr) {
    char* w = static_cast<char*>(ws);
    float* hit_rec = reinterpret_cast<float*>(w + op.off_list);
    int* hit_count = reinterpret_cast<int*>(w + op.off_count);
    unsigned int* gmax_bits = reinterpret_cast<unsigned int*>(w + op.off_gmax);
    FBBEV_LAUNCH(k_da_bwd_init, 1, 256, 0, stream, B * Ncam, hit_count, B, gmax_bits);
    FBBEV_CHECK_LAUNCH();
    int e = 0;
    // (the plane kernel reads a record's 4 mask bytes / 8 reference floats / 4 depths as whole words: alignment of the geometry inputs)
    if (op.unit_planes && Za == FBBEV_DAF_ZA && aligned16(ref_cam) && aligned16(qdepth) && ((uintptr_t)mask & 3) == 0) {
        // camera tokens as head planes, then the unit gradients with the forward's (head, patch) mapping
        const float* planes = planes_in;
        if (!planes) {                                                 // (round 6: the training forward's own head planes may be handed in)
            float* pl_w = reinterpret_cast<float*>(w + op.off_planes);
            const long long n_el = (long long)B * Ncam * S * M * Dh;
            FBBEV_LAUNCH(k_value_rows_to_head_planes, (n_el + 255) / 256, 256, 0, stream, value, (long long)B * Ncam * S, S, M, Dh, HS,
                         (head_minor & 4) ? 1 : 0, pl_w);
            FBBEV_CHECK_LAUNCH();
            planes = pl_w;
        }
        const int gw = (bev_w > 0 && Q % bev_w == 0) ? bev_w : 0;
        const long long wgs_u = gw > 0 ? (long long)B * ((gw + 7) / 8) * ((Q / gw + 7) / 8) : (long long)B * ((Q + 63) / 64);
        const long long grid_u = (wgs_u + 7) / 8 * 8;
        const size_t lds_u = fbbev_dbp_lds_bytes(M, Ncam);
        if (grid_u >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
#define FBBEV_DA_UNIT_PLANES(DH_)                                                                                     \
    do {                                                                                                               \
        e = fbbev_rt_allow_dyn_lds((const void*)k_da_bwd_unit_planes<DH_, 8>, lds_u);                                  \
        if (e) return e;                                                                                               \
        FBBEV_LAUNCH((k_da_bwd_unit_planes<DH_, 8>), grid_u, 512, lds_u, stream, (const float*)planes, spatial_shapes,  \
                     level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots, B, Ncam, S, L, Q, \
                     gw, DC, d0, dstep, head_minor & 3, grad_pred_depth, grad_offsets, grad_attn, gmax_bits);           \
    } while (0)
        if (Dh == 10) FBBEV_DA_UNIT_PLANES(10); else FBBEV_DA_UNIT_PLANES(8);
#undef FBBEV_DA_UNIT_PLANES
        FBBEV_CHECK_LAUNCH();
    } else {
        e = da_bwd_unit_launch(stream, value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn,
                               grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, HS, grad_pred_depth,
                               grad_offsets, grad_attn, gmax_bits);
        if (e) return e;
    }
    FBBEV_LAUNCH(k_da_bwd_hitlist, (long long)B * ((Q + 255) / 256), 256, 0, stream, spatial_shapes, pred_depth, ref_cam, mask,
                 qdepth, B, Ncam, Q, Za, DC, d0, dstep, hit_rec, hit_count);
    FBBEV_CHECK_LAUNCH();
    const fbbev_da_bwd_region_tab& tab = op.tab;
    const long long wgs = (long long)tab.n * B * Ncam * M;
#define FBBEV_DA_OWN(NT_, DH_)                                                                                          \
    do {                                                                                                                \
        e = fbbev_rt_allow_dyn_lds((const void*)k_da_bwd_scatter_owned<NT_, DH_>, op.lds);                              \
        if (e) return e;                                                                                                \
        FBBEV_LAUNCH((k_da_bwd_scatter_owned<NT_, DH_>), wgs, NT_, op.lds, stream, spatial_shapes, level_start_index,   \
                     ref_cam, offsets, attn, grad_slots, B, Ncam, S, M, L, Q, P, Za, head_minor & 3, HS, tab,            \
                     (const float*)hit_rec, (const int*)hit_count,                                                       \
                     (const unsigned int*)gmax_bits, (head_minor & 4) ? 1 : 0, grad_value);                              \
    } while (0)
#define FBBEV_DA_OWN_NT(DH_) do { if (op.threads == 512) FBBEV_DA_OWN(512, DH_); else FBBEV_DA_OWN(256, DH_); } while (0)   /* 1024 threads measured slower */
    if (Dh == 10) FBBEV_DA_OWN_NT(10);
    else if (Dh == 8) FBBEV_DA_OWN_NT(8);
    else if (Dh == 4) FBBEV_DA_OWN_NT(4);
    else FBBEV_DA_OWN_NT(16);
#undef FBBEV_DA_OWN_NT
#undef FBBEV_DA_OWN
    FBBEV_CHECK_LAUNCH();
    return 0;
}

static int da_cross_attn_bwd_ws_impl(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                                          const float* qdepth, const float* offsets, const float* attn,
                                          const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                                          int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                                          float* grad_value, float* grad_pred_depth, float* grad_offsets,
                                          float* grad_attn, const int32_t* level_hw_host, void* ws, size_t ws_bytes,
                                          fbbev_stream_t stream_, int bev_w) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q < 0 || P <= 0 || Za <= 0 || DC <= 0 || bev_w < 0)
        return FBBEV_E_BADARG;
    const int HS = head_stride == 0 ? Dh : head_stride;
    {
        da_own_plan op;
        if (HS >= Dh && Q > 0 && ws && aligned16(ws) && aligned16(grad_value) && value && spatial_shapes && level_start_index &&
            pred_depth && ref_cam && mask && qdepth && offsets && attn && grad_slots && grad_value && grad_pred_depth &&
            grad_offsets && grad_attn && dstep != 0.f && P % Za == 0 &&
            (((uintptr_t)offsets | (uintptr_t)grad_offsets | (uintptr_t)grad_slots) & 7) == 0 &&
            da_own_plan_make(B, Ncam, S, M, Dh, Q, HS, L, P, Za, level_hw_host, &op) && ws_bytes >= op.ws)
            return da_bwd_owned_launch(op, (fbbev_rt_stream)stream_, value, spatial_shapes, level_start_index, pred_depth, ref_cam,
                                       mask, qdepth, offsets, attn, grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep,
                                       head_minor, HS, grad_value, grad_pred_depth, grad_offsets, grad_attn, ws, bev_w);
    }
    da_bwd_plan pl;
    if (HS < Dh || Q == 0 || !ws || !aligned16(ws) || !aligned16(grad_value) || !aligned16(value) ||
        !da_bwd_tile_plan(B, Ncam, S, M, Dh, Q, HS, L, P, level_hw_host, &pl) || ws_bytes < pl.ws ||
        (long long)B * M * pl.chunks >= (1ll << 31) || (long long)B * Ncam * S * M * HS * 4 >= (1ll << 32) ||
        (((uintptr_t)offsets | (uintptr_t)grad_offsets | (uintptr_t)grad_slots) & 7) != 0)
        return fbbev_da_cross_attn_bwd(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                                       attn, grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor,
                                       head_stride, grad_value, grad_pred_depth, grad_offsets, grad_attn, stream_);
    if (Za > FBBEV_DA_MAX_ZA || P % Za != 0) return FBBEV_E_UNSUPPORTED;
    if (dstep == 0.f) return FBBEV_E_BADARG;
    if (!value || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth || !offsets ||
        !attn || !grad_slots || !grad_value || !grad_pred_depth || !grad_offsets || !grad_attn) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    float* part = static_cast<float*>(ws);
    const long long wgs = (long long)B * M * pl.chunks;
    const float* info = nullptr;
    if (pl.info_stride > 0 && 1 + Za <= pl.info_stride) {
        float* info_w = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.ws_part);
        FBBEV_LAUNCH(k_da_bwd_hitinfo, ((long long)B * Q + 255) / 256, 256, 0, stream, spatial_shapes, pred_depth, ref_cam, mask,
                     qdepth, B, Ncam, Q, Za, DC, d0, dstep, pl.info_stride, info_w);
        FBBEV_CHECK_LAUNCH();
        info = info_w;
    }
    {
        {
            const int e = da_bwd_unit_launch(stream, value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets,
                                             attn, grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, HS,
                                             grad_pred_depth, grad_offsets, grad_attn, nullptr);
            if (e) return e;
        }
        // (B) value gradient, one launch per token region
        for (int r = 0; r < pl.n_regions; ++r) {
            const da_region& rg = pl.reg[r];
            // small regions (coarse levels): up to 4 copies of the plane inside the LDS budget of the largest region
            const size_t one = (size_t)FBBEV_DA_PLANE_WORDS(rg.tok1 - rg.tok0, HS) * sizeof(long long);
            int copies = (int)((size_t)(da_bwd_plane_kb() * 1024) / one);
            copies = copies < 1 ? 1 : (copies > 4 ? 4 : copies);
            { const int v = da_bwd_env().copies; if (v >= 1 && v <= copies) copies = v; }
            const size_t lds_b = one * copies + (size_t)((pl.q_per_chunk + 1) & ~1) * 2 + (size_t)(1 + pl.threads / 64) * sizeof(int);
#define FBBEV_DA_BWD_SC(NT_, DH_)                                                                                       \
    FBBEV_LAUNCH((k_da_cross_attn_bwd_scatter<NT_, DH_>), wgs, NT_, lds_b, stream, spatial_shapes, level_start_index,     \
                 pred_depth, ref_cam, mask, qdepth, offsets, attn, grad_slots, B, Ncam, S, M, L, Q, P, Za, DC, d0, dstep, \
                 head_minor & 3, HS, pl.chunks, pl.q_per_chunk, rg.lvl0, rg.lvl1, rg.tok0, rg.tok1, copies, part, info,  \
                 pl.info_stride)
#define FBBEV_DA_BWD_SC_NT(DH_) do { if (pl.threads == 512) FBBEV_DA_BWD_SC(512, DH_); else FBBEV_DA_BWD_SC(256, DH_); } while (0)
            if (Dh == 10) FBBEV_DA_BWD_SC_NT(10);
            else if (Dh == 8) FBBEV_DA_BWD_SC_NT(8);
            else if (Dh == 4) FBBEV_DA_BWD_SC_NT(4);
            else FBBEV_DA_BWD_SC_NT(16);
#undef FBBEV_DA_BWD_SC_NT
#undef FBBEV_DA_BWD_SC
            FBBEV_CHECK_LAUNCH();
        }
    }
    const long long n = (long long)B * Ncam * S * M * HS;
    long long rb = (n + 255) / 256;
    if (rb > 65536) rb = 65536;
    FBBEV_LAUNCH(k_da_bwd_reduce, rb, 256, 0, stream, (const float*)part, B, Ncam, S, M, HS, pl.chunks, (head_minor & 4) ? 1 : 0,
                 grad_value);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_da_cross_attn_bwd_ws(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                                          const float* qdepth, const float* offsets, const float* attn,
                                          const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                                          int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                                          float* grad_value, float* grad_pred_depth, float* grad_offsets,
                                          float* grad_attn, const int32_t* level_hw_host, void* ws, size_t ws_bytes,
                                          fbbev_stream_t stream_) {
    return da_cross_attn_bwd_ws_impl(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn,
                                     grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, head_stride,
                                     grad_value, grad_pred_depth, grad_offsets, grad_attn, level_hw_host, ws, ws_bytes, stream_, 0);
}
// ... with the BEV grid's width: the queries are a (Q / bev_w) x bev_w grid, so the unit gradients take 8 x 8 patches of it
// (k_da_bwd_unit_planes; bev_w = 0 or not a divisor of Q: runs of 64 consecutive queries)
extern "C" int fbbev_da_cross_attn_bwd_ws_grid(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                               const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                                               const float* qdepth, const float* offsets, const float* attn,
                                               const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                                               int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                                               float* grad_value, float* grad_pred_depth, float* grad_offsets,
                                               float* grad_attn, const int32_t* level_hw_host, void* ws, size_t ws_bytes,
                                               int bev_w, fbbev_stream_t stream_) {
    return da_cross_attn_bwd_ws_impl(value, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth, offsets, attn,
                                     grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, head_stride,
                                     grad_value, grad_pred_depth, grad_offsets, grad_attn, level_hw_host, ws, ws_bytes, stream_,
                                     bev_w);
}

// Round 6: the backward of fbbev_da_cross_attn_fused for the training step that runs the one-kernel forward -- the camera tokens arrive as
// the HEAD PLANES the forward sampled (fbbev_rows_linear_x3_planes: no row copy of them exists), and on this route every output but
// grad_pred_depth is WRITTEN in full (grad_value by the plane owners incl. the padding channels, grad_offsets / grad_attn by the
// unit kernel for every query of the grid): the caller zeroes only grad_pred_depth.  Needs the output-owned plane route with the
// unit gradients on head planes (M = 8, Dh in {8, 10}, 8 points, 4 anchors, levels >= 2 wide, >= 256 planes) and queries on a
// bev_w-wide grid; FBBEV_E_UNSUPPORTED otherwise (the caller keeps fbbev_da_cross_attn_bwd_ws_grid on row tokens).
extern "C" int fbbev_da_cross_attn_bwd_planes_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int head_stride,
                                                        const int32_t* level_hw_host, int bev_w) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0 || Za <= 0 || bev_w <= 0 || Q % bev_w != 0) return 0;
    const int HS = head_stride == 0 ? Dh : head_stride;
    da_own_plan op;
    if (HS < Dh || P % Za != 0 || Za != FBBEV_DAF_ZA || !da_own_plan_make(B, Ncam, S, M, Dh, Q, HS, L, P, Za, level_hw_host, &op)) return 0;
    return op.unit_planes ? 1 : 0;
}
extern "C" int fbbev_da_cross_attn_bwd_planes(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                              const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                                              const float* qdepth, const float* offsets, const float* attn, const float* grad_slots,
                                              int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0,
                                              float dstep, int head_minor, int head_stride, float* grad_value, float* grad_pred_depth,
                                              float* grad_offsets, float* grad_attn, const int32_t* level_hw_host, void* ws,
                                              size_t ws_bytes, int bev_w, fbbev_stream_t stream_) {
    if (B <= 0 || Ncam <= 0 || S <= 0 || M <= 0 || Dh <= 0 || L <= 0 || Q <= 0 || P <= 0 || Za <= 0 || DC <= 0 || bev_w <= 0)
        return FBBEV_E_BADARG;
    if (!planes || !spatial_shapes || !level_start_index || !pred_depth || !ref_cam || !mask || !qdepth || !offsets || !attn ||
        !grad_slots || !grad_value || !grad_pred_depth || !grad_offsets || !grad_attn || !ws) return FBBEV_E_BADARG;
    if (dstep == 0.f) return FBBEV_E_BADARG;
    const int HS = head_stride == 0 ? Dh : head_stride;
    da_own_plan op;
    if (HS < Dh || P % Za != 0 || Za != FBBEV_DAF_ZA || Q % bev_w != 0 ||
        !da_own_plan_make(B, Ncam, S, M, Dh, Q, HS, L, P, Za, level_hw_host, &op) || !op.unit_planes) return FBBEV_E_UNSUPPORTED;
    if (!aligned16(ws) || !aligned16(grad_value) || !aligned16(ref_cam) || !aligned16(qdepth) || ((uintptr_t)mask & 3) != 0 ||
        ((uintptr_t)planes & 7) != 0 || (((uintptr_t)offsets | (uintptr_t)grad_offsets | (uintptr_t)grad_slots) & 7) != 0)
        return FBBEV_E_UNSUPPORTED;
    if (ws_bytes < op.ws) return FBBEV_E_WORKSPACE;
    return da_bwd_owned_launch(op, (fbbev_rt_stream)stream_, nullptr, spatial_shapes, level_start_index, pred_depth, ref_cam, mask, qdepth,
                               offsets, attn, grad_slots, B, Ncam, S, M, Dh, L, Q, P, Za, DC, d0, dstep, head_minor, HS, grad_value,
                               grad_pred_depth, grad_offsets, grad_attn, ws, bev_w, planes);
}

// ---------------------------------------------------------------- fused lift-splat backward (training)
struct bwd_layout { size_t table, meta, rows, total; long long n_tiles; int tpp; long long max_rows; };

static bwd_layout pool_bwd_layout(int B, int N, int D, int H, int W, int C, int Z, int Y, int X) {
    bwd_layout l;
    const long long n = (long long)B * N * D * H * W, yx = (long long)Y * X, nvox = (long long)B * Z * yx;
    l.tpp = (int)((yx + 127) / 128);
    l.n_tiles = (long long)B * Z * l.tpp;
    l.max_rows = n < nvox ? n : nvox;                 // I <= min(points, voxels)
    l.table = 0;
    l.meta = align_up((size_t)n * 4, 256);
    l.rows = l.meta + align_up((size_t)(l.n_tiles + 1) * 8, 256);
    l.total = l.rows + align_up((size_t)l.max_rows * C * 4, 256);
    return l;
}

extern "C" size_t fbbev_pool_dense_bwd_workspace_bytes(int B, int N, int D, int H, int W, int C, int Z, int Y, int X) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 256;
    return pool_bwd_layout(B, N, D, H, W, C, Z, Y, X).total;
}

static int pool_dense_bwd_impl(const float* out_grad, long long og_stride_b, long long og_stride_c,
                               const float* depth, const float* feat, const int32_t* ranks_depth,
                               const int32_t* interval_rank, const int32_t* interval_starts,
                               const int32_t* counts, int n_intervals_max, int B, int N, int D,
                               int H, int W, int C, int Z, int Y, int X, float* depth_grad,
                               float* feat_grad, void* workspace, size_t workspace_bytes,
                               fbbev_stream_t stream_, const float* zgrad, float zscale) {
    if (zgrad && !aligned16(zgrad)) return FBBEV_E_UNSUPPORTED;
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Z <= 0 || Y <= 0 || X <= 0 || n_intervals_max < 0)
        return FBBEV_E_BADARG;
    if (!out_grad || !depth || !feat || !ranks_depth || !interval_rank || !interval_starts || !counts ||
        !depth_grad || !feat_grad || !workspace) return FBBEV_E_BADARG;
    const long long yx = (long long)Y * X, n = (long long)B * N * D * H * W;
    if (C % 4 != 0 || C > 256 || yx % 4 != 0 || !aligned16(out_grad) || !aligned16(feat) || !aligned16(feat_grad) ||
        !aligned16(workspace)) return FBBEV_E_UNSUPPORTED;
    if ((long long)B * Z * yx >= (1ll << 31) || n >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (og_stride_c == 0) og_stride_c = (long long)Z * yx;
    if (og_stride_b == 0) og_stride_b = (long long)C * og_stride_c;
    if (og_stride_c < (long long)Z * yx || og_stride_b < (long long)C * og_stride_c || og_stride_c % 4 != 0 ||
        og_stride_b % 4 != 0) return FBBEV_E_BADARG;
    const bwd_layout l = pool_bwd_layout(B, N, D, H, W, C, Z, Y, X);
    if (workspace_bytes < l.total) return FBBEV_E_WORKSPACE;
    const size_t lds = (size_t)C * (128 + 4) * 4;
    if (lds > 160 * 1024) return FBBEV_E_UNSUPPORTED;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    char* ws = static_cast<char*>(workspace);
    int* table = reinterpret_cast<int*>(ws + l.table);
    int* meta = reinterpret_cast<int*>(ws + l.meta);
    float* rows = reinterpret_cast<float*>(ws + l.rows);
    int e = fbbev_rt_memset_async(table, 0xFF, (size_t)n * 4, stream);      // -1 = point dropped
    if (e) return e;
    {
        long long blocks = (n + 255) / 256;      // P <= n, grid-stride over the device-side P
        if (blocks > 8192) blocks = 8192;
        FBBEV_LAUNCH(k_point_row_table, blocks, 256, 0, stream, ranks_depth, interval_starts, counts,
                     n_intervals_max, table);
        FBBEV_CHECK_LAUNCH();
    }
    FBBEV_LAUNCH(k_tile_lower_bound2, (l.n_tiles + 1 + 255) / 256, 256, 0, stream, (int)l.n_tiles, l.tpp, (int)yx,
                 128, interval_rank, interval_starts, counts, n_intervals_max, (const int*)nullptr, meta);
    FBBEV_CHECK_LAUNCH();
    if (lds > 64 * 1024) {
        e = fbbev_rt_allow_dyn_lds((const void*)k_pool_bwd_rows<128>, lds);
        if (e) return e;
    }
    FBBEV_LAUNCH(k_pool_bwd_rows<128>, l.n_tiles, 256, lds, stream, C, Z, (int)yx, l.tpp, og_stride_b, og_stride_c,
                 out_grad, interval_rank, meta, rows, zgrad, zscale);
    FBBEV_CHECK_LAUNCH();
    const long long n_pixels = (long long)B * N * H * W;
    const long long blocks = (n_pixels + 7) / 8;      // 8 half-waves per 256-thread workgroup
    if (C <= 128)
        FBBEV_LAUNCH(k_pool_bwd_pixel<4>, blocks, 256, 0, stream, C, D, H * W, n_pixels, (long long)C, rows, depth,
                     feat, table, depth_grad, feat_grad);
    else if (C % 8 == 0)
        FBBEV_LAUNCH(k_pool_bwd_pixel<8>, blocks, 256, 0, stream, C, D, H * W, n_pixels, (long long)C, rows, depth,
                     feat, table, depth_grad, feat_grad);
    else
        return FBBEV_E_UNSUPPORTED;
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_bev_pool_v2_dense_bwd(const float* out_grad, long long og_stride_b, long long og_stride_c,
                                           const float* depth, const float* feat, const int32_t* ranks_depth,
                                           const int32_t* interval_rank, const int32_t* interval_starts,
                                           const int32_t* counts, int n_intervals_max, int B, int N, int D,
                                           int H, int W, int C, int Z, int Y, int X, float* depth_grad,
                                           float* feat_grad, void* workspace, size_t workspace_bytes,
                                           fbbev_stream_t stream_) {
    return pool_dense_bwd_impl(out_grad, og_stride_b, og_stride_c, depth, feat, ranks_depth, interval_rank, interval_starts, counts,
                               n_intervals_max, B, N, D, H, W, C, Z, Y, X, depth_grad, feat_grad, workspace, workspace_bytes, stream_,
                               nullptr, 0.f);
}
// ... with a second gradient (B, C, Y, X) that every z plane receives, scaled by zscale: out_grad_eff = out_grad + zscale * zgrad[..., None]
extern "C" int fbbev_bev_pool_v2_dense_bwd_z(const float* out_grad, long long og_stride_b, long long og_stride_c,
                                             const float* zgrad, float zscale, const float* depth, const float* feat,
                                             const int32_t* ranks_depth, const int32_t* interval_rank,
                                             const int32_t* interval_starts, const int32_t* counts, int n_intervals_max, int B,
                                             int N, int D, int H, int W, int C, int Z, int Y, int X, float* depth_grad,
                                             float* feat_grad, void* workspace, size_t workspace_bytes, fbbev_stream_t stream_) {
    if (!zgrad) return FBBEV_E_BADARG;
    return pool_dense_bwd_impl(out_grad, og_stride_b, og_stride_c, depth, feat, ranks_depth, interval_rank, interval_starts, counts,
                               n_intervals_max, B, N, D, H, W, C, Z, Y, X, depth_grad, feat_grad, workspace, workspace_bytes, stream_,
                               zgrad, zscale);
}

// ------------------------------------------------------------------------------ temporal history alignment
extern "C" int fbbev_history_flow(const float* history_forward_augs, const float* curr_to_prev_ego_rt, const float* bda,
                                  const float* dx3, const float* lower3, int B, float* rt_flow, fbbev_stream_t stream_) {
    if (B < 0) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!history_forward_augs || !curr_to_prev_ego_rt || !bda || !dx3 || !lower3 || !rt_flow) return FBBEV_E_BADARG;
    if (!(dx3[0] > 0.f) || !(dx3[1] > 0.f) || !(dx3[2] > 0.f)) return FBBEV_E_BADARG;
    FBBEV_LAUNCH(k_history_flow, (B + 63) / 64, 64, 0, (fbbev_rt_stream)stream_, history_forward_augs, curr_to_prev_ego_rt,
                 bda, dx3[0], dx3[1], dx3[2], lower3[0], lower3[1], lower3[2], B, rt_flow);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_warp_e(const void* history, long long history_stride_b, const float* rt_flow, int B, int CH,
                                    int Z, int Y, int X, void* out, long long out_stride_b, int elem_type,
                                    fbbev_stream_t stream_) {
    if (B < 0 || CH < 0 || Z < 2 || Y < 2 || X < 2) return FBBEV_E_BADARG;      // size-1 axes divide by zero in :208
    if (elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (B == 0 || CH == 0) return 0;
    if (!history || !rt_flow || !out) return FBBEV_E_BADARG;
    const long long zyx = (long long)Z * Y * X;
    if (zyx >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (history_stride_b == 0) history_stride_b = (long long)CH * zyx;
    if (out_stride_b == 0) out_stride_b = (long long)CH * zyx;
    if (history_stride_b < (long long)CH * zyx || out_stride_b < (long long)CH * zyx) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    // FBBEV_HISTORY_WARP=lds selects the LDS-staged brick kernel (k_history_warp_lds: bit-identical, but as built --
    // one 1024-thread workgroup per CU, staging and taps of a channel serialised by two barriers -- 3x SLOWER than the
    // gather kernel on MI355X, profiles/r02_time_history_lds_vs_gather.jsonl; kept for the next iteration, not the default)
    const char* mode = getenv("FBBEV_HISTORY_WARP");
    const bool lds = mode && mode[0] == 'l';
    if (lds && zyx >= 4096 && X >= 16) {
        int TX = 64;
        while (TX / 2 >= X) TX /= 2;
        const int BZ = Z < 8 ? Z : 8;
        int TY = 4096 / (BZ * TX);
        if (TY > Y) TY = Y;
        if (TY < 1) TY = 1;
        const int ntx = (X + TX - 1) / TX, nty = (Y + TY - 1) / TY, ntz = (Z + BZ - 1) / BZ;
        int cpb = 64;                                  // channels per workgroup: the brick's tap setup is amortised over them
        while (cpb > 8 && (long long)B * ntz * nty * ntx * ((CH + cpb - 1) / cpb) < 2048) cpb >>= 1;
        const int n_groups = (CH + cpb - 1) / cpb;
        const long long blocks = (long long)B * n_groups * ntz * nty * ntx;
        if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
        if (elem_type == 0)
            FBBEV_LAUNCH(k_history_warp_lds<0>, blocks, 1024, 0, stream, history, history_stride_b, rt_flow, CH, Z, Y, X, BZ, TY,
                         TX, ntz, nty, ntx, cpb, n_groups, out, out_stride_b);
        else if (elem_type == 1)
            FBBEV_LAUNCH(k_history_warp_lds<1>, blocks, 1024, 0, stream, history, history_stride_b, rt_flow, CH, Z, Y, X, BZ, TY,
                         TX, ntz, nty, ntx, cpb, n_groups, out, out_stride_b);
        else
            FBBEV_LAUNCH(k_history_warp_lds<2>, blocks, 1024, 0, stream, history, history_stride_b, rt_flow, CH, Z, Y, X, BZ, TY,
                         TX, ntz, nty, ntx, cpb, n_groups, out, out_stride_b);
        FBBEV_CHECK_LAUNCH();
        return 0;
    }
    const int n_chunks = (int)((zyx + 255) / 256);
    // enough workgroups to fill 256 CUs several times over, at least 8 channels each to amortise the tap setup
    int cpb = 64;
    while (cpb > 8 && (long long)B * n_chunks * ((CH + cpb - 1) / cpb) < 4096) cpb >>= 1;
    const int n_groups = (CH + cpb - 1) / cpb;
    const long long blocks = (long long)B * n_groups * n_chunks;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int per_xcd = (int)((blocks + 7) / 8);
    if (elem_type == 0)
        FBBEV_LAUNCH(k_history_warp<0>, (long long)per_xcd * 8, 256, 0, stream, history, history_stride_b, rt_flow,
                     CH, Z, Y, X, cpb, n_groups, n_chunks, per_xcd, (int)blocks, out, out_stride_b);
    else if (elem_type == 1)
        FBBEV_LAUNCH(k_history_warp<1>, (long long)per_xcd * 8, 256, 0, stream, history, history_stride_b, rt_flow,
                     CH, Z, Y, X, cpb, n_groups, n_chunks, per_xcd, (int)blocks, out, out_stride_b);
    else
        FBBEV_LAUNCH(k_history_warp<2>, (long long)per_xcd * 8, 256, 0, stream, history, history_stride_b, rt_flow,
                     CH, Z, Y, X, cpb, n_groups, n_chunks, per_xcd, (int)blocks, out, out_stride_b);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_warp(const float* history, long long history_stride_b, const float* rt_flow, int B, int CH,
                                  int Z, int Y, int X, float* out, long long out_stride_b, fbbev_stream_t stream_) {
    return fbbev_history_warp_e(history, history_stride_b, rt_flow, B, CH, Z, Y, X, out, out_stride_b, 0, stream_);
}

// voxel-major ring: frames [T][N][C] per sample (history_kernels.h)
struct fbbev_warp_vm_plan { int groups, n_xc, YB, nyb; };

static int history_warp_vm_plan(const void* history, long long& history_stride_b, const float* rt_flow, int B, int T, int C,
                                int Z, int Y, int X, void* out, long long& out_stride_b, int elem_type, fbbev_warp_vm_plan& pl) {
    if (B < 0 || T < 0 || C <= 0 || Z < 2 || Y < 2 || X < 2) return FBBEV_E_BADARG;
    if (elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (B == 0 || T == 0) return 0;
    if (!history || !rt_flow || !out) return FBBEV_E_BADARG;
    const int VE = elem_type == 0 ? 4 : 8;
    const long long zyx = (long long)Z * Y * X, frame = zyx * C;
    if (zyx >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (history_stride_b == 0) history_stride_b = (long long)T * frame;
    if (out_stride_b == 0) out_stride_b = (long long)T * frame;
    if (history_stride_b < (long long)T * frame || out_stride_b < (long long)T * frame) return FBBEV_E_BADARG;
    if (C % VE != 0 || history_stride_b % VE != 0 || out_stride_b % VE != 0 || !aligned16(history) || !aligned16(out))
        return FBBEV_E_UNSUPPORTED;
    pl.groups = C / VE;
    if (frame * (elem_type == 0 ? 4 : 2) >= (1ll << 32)) return FBBEV_E_UNSUPPORTED;      // 32-bit byte offsets inside a frame
    pl.n_xc = (X * pl.groups + 255) / 256;                        // workgroups per grid row
    int YB = 128 / Z;                                             // rows per band: a (z, y) slab of ~128 workgroups per x chunk
    if (const char* e = getenv("FBBEV_HISTORY_VM_YB")) YB = atoi(e);        // tuning knob (profiles/r02_time_history_bf16_voxel_major.jsonl)
    if (YB < 1) YB = 1;
    if (YB > Y) YB = Y;
    pl.YB = YB;
    pl.nyb = (Y + YB - 1) / YB;
    return 0;
}

// the bands yb0 .. yb0 + nyb_c - 1 of every sample (all of them: the whole warp)
static int history_warp_vm_bands(const void* history, long long history_stride_b, const float* rt_flow, int B, int T, int C,
                                 int Z, int Y, int X, void* out, long long out_stride_b, int elem_type,
                                 const fbbev_warp_vm_plan& pl, int yb0, int nyb_c, fbbev_rt_stream stream) {
    constexpr int TU = 2;
    const int groups = pl.groups, n_xc = pl.n_xc, YB = pl.YB;
    const long long blocks = (long long)B * nyb_c * n_xc * YB * Z;
    if (blocks >= (1ll << 31) - 8) return FBBEV_E_UNSUPPORTED;
    if (blocks == 0) return 0;
    const int per_xcd = (int)((blocks + 7) / 8);
    // occupancy experiment knob (round 3, profiles/r03_exp_history_occupancy.jsonl): dynamic LDS the kernel does not use, in KB
    static const size_t warp_pad = [] { const char* e = getenv("FBBEV_HISTORY_WARP_LDS_PAD_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
    if (warp_pad > 64 * 1024) {
        fbbev_rt_allow_dyn_lds((const void*)k_history_warp_vm<0, TU, 1>, warp_pad);
        fbbev_rt_allow_dyn_lds((const void*)k_history_warp_vm<1, TU, 1>, warp_pad);
        fbbev_rt_allow_dyn_lds((const void*)k_history_warp_vm<2, TU, 1>, warp_pad);
        fbbev_rt_allow_dyn_lds((const void*)k_history_warp_vm<2, 4, 1>, warp_pad);
    }
#define FBBEV_HWVM(ET_, TU_, ST_)                                                                                            \
    FBBEV_LAUNCH((k_history_warp_vm<ET_, TU_, ST_>), (long long)per_xcd * 8, 256, warp_pad, stream, history, history_stride_b, rt_flow, \
                 T, C, Z, Y, X, groups, n_xc, YB, nyb_c, per_xcd, (int)blocks, out, out_stride_b, yb0)
    // ST = 1: non-temporal ring stores (A/B on one box: 4.0 vs 4.2 ms at 400x400x16, 0.56 vs 0.61 ms at 100x100x8 B=4)
    // fp16 ring: four frames (32 taps) in flight per thread -- the widening rides in the multiply (v_fma_mix_f32), which left the
    // registers for it (round 6: 3.45 -> 3.33 ms at 400x400x16); FBBEV_HISTORY_VM_TU=2 restores two
    static const int tu2 = [] { const char* e = getenv("FBBEV_HISTORY_VM_TU"); return e && atoi(e) == 2 ? 1 : 0; }();
    if (elem_type == 0) FBBEV_HWVM(0, TU, 1);
    else if (elem_type == 1) FBBEV_HWVM(1, TU, 1);
    else if (tu2) FBBEV_HWVM(2, TU, 1);
    else FBBEV_HWVM(2, 4, 1);
#undef FBBEV_HWVM
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_warp_vm(const void* history, long long history_stride_b, const float* rt_flow, int B, int T, int C,
                                     int Z, int Y, int X, void* out, long long out_stride_b, int elem_type,
                                     fbbev_stream_t stream_) {
    fbbev_warp_vm_plan pl{};
    const int e = history_warp_vm_plan(history, history_stride_b, rt_flow, B, T, C, Z, Y, X, out, out_stride_b, elem_type, pl);
    if (e || B == 0 || T == 0) return e;
    return history_warp_vm_bands(history, history_stride_b, rt_flow, B, T, C, Z, Y, X, out, out_stride_b, elem_type, pl, 0, pl.nyb,
                                 (fbbev_rt_stream)stream_);
}

extern "C" int fbbev_history_frame_vm(const float* curr, int B, int C, int N, int inner, void* out, long long out_stride_b,
                                      int elem_type, fbbev_stream_t stream_) {
    if (B < 0 || C <= 0 || N < 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (inner < 1 || N % inner != 0) return FBBEV_E_BADARG;
    if (B == 0 || N == 0) return 0;
    if (!curr || !out) return FBBEV_E_BADARG;
    const int VE = elem_type == 0 ? 4 : 8;
    if (out_stride_b == 0) out_stride_b = (long long)N * C;
    if (out_stride_b < (long long)N * C) return FBBEV_E_BADARG;
    if (C % VE != 0 || C > 512 || out_stride_b % VE != 0 || !aligned16(out)) return FBBEV_E_UNSUPPORTED;
    const int tiles_per_b = (N + 63) / 64;
    const long long blocks = (long long)B * tiles_per_b;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = (size_t)C * 65 * sizeof(float);
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    if (lds > 64 * 1024) {
        const void* k = elem_type == 0 ? (const void*)k_history_frame_vm<0> : elem_type == 1 ? (const void*)k_history_frame_vm<1> : (const void*)k_history_frame_vm<2>;
        const int e = fbbev_rt_allow_dyn_lds(k, lds);
        if (e) return e;
    }
    if (elem_type == 0) FBBEV_LAUNCH(k_history_frame_vm<0>, blocks, 256, lds, stream, curr, C, N, inner, tiles_per_b, out, out_stride_b);
    else if (elem_type == 1) FBBEV_LAUNCH(k_history_frame_vm<1>, blocks, 256, lds, stream, curr, C, N, inner, tiles_per_b, out, out_stride_b);
    else FBBEV_LAUNCH(k_history_frame_vm<2>, blocks, 256, lds, stream, curr, C, N, inner, tiles_per_b, out, out_stride_b);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ LayerNorm over short rows
extern "C" int fbbev_layernorm(const float* x, const float* residual, const float* weight, const float* bias, float eps,
                               long long rows, int C, float* out, fbbev_stream_t stream_) {
    if (rows < 0 || C <= 0 || !(eps >= 0.f)) return FBBEV_E_BADARG;
    if (rows == 0) return 0;
    if (!x || !weight || !bias || !out) return FBBEV_E_BADARG;
    if (C % 4 != 0 || C > 128 || !aligned16(x) || !aligned16(out) || !aligned16(weight) || !aligned16(bias) ||
        (residual && !aligned16(residual))) return FBBEV_E_UNSUPPORTED;
    const long long blocks = (rows + 7) / 8;          // 8 half-waves per 256-thread workgroup
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_layernorm_rows, blocks, 256, 0, (fbbev_rt_stream)stream_, x, residual, weight, bias, eps, rows, C, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_layernorm_bwd_partials(long long rows) {
    // workgroups of fbbev_layernorm_bwd = rows of its `partial` output: enough to fill the chip, few enough that the
    // final sum over them stays small
    long long wgs = (rows + 7) / 8;
    if (wgs > 2048) wgs = 2048;
    return (int)(wgs < 1 ? 1 : wgs);
}

extern "C" int fbbev_layernorm_bwd(const float* x, const float* grad_out, const float* weight, float eps, long long rows,
                                   int C, float* grad_x, float* partial, fbbev_stream_t stream_) {
    if (rows < 0 || C <= 0 || !(eps >= 0.f)) return FBBEV_E_BADARG;
    if (!partial) return FBBEV_E_BADARG;
    if (rows > 0 && (!x || !grad_out || !weight || !grad_x)) return FBBEV_E_BADARG;
    if (C % 4 != 0 || C > 128 || !aligned16(x) || !aligned16(grad_out) || !aligned16(weight) || !aligned16(grad_x))
        return FBBEV_E_UNSUPPORTED;
    const int wgs = fbbev_layernorm_bwd_partials(rows);
    FBBEV_LAUNCH(k_layernorm_rows_bwd, wgs, 256, 0, (fbbev_rt_stream)stream_, x, grad_out, weight, eps, rows, C, grad_x, partial);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

template <int ET, bool VM = false>
static int history_conv_launch(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                               const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                               float* out, void* workspace, size_t workspace_bytes, fbbev_rt_stream stream) {
    const int tiles_per_b = (N + 63) / 64;
    const long long blocks = (long long)B * tiles_per_b;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = (size_t)4 * C * 16 * sizeof(float);
    if (workspace && ((C == 80 && Cout == 80) || (C == 16 && Cout == 16))) {
        // fragment-ordered copies of the two weight matrices (tiny: (1 + T1) * C * C floats), then the register-resident kernel
        const int MT1 = C / 16, MT2 = Cout / 16, KS = C / 4;
        const size_t need = ((size_t)MT1 * KS + (size_t)T1 * MT2 * KS) * 64 * sizeof(float);
        if (workspace_bytes < need) return FBBEV_E_WORKSPACE;
        float* w1f = static_cast<float*>(workspace);
        float* w2f = w1f + (size_t)MT1 * KS * 64;
        const int nfrag = (MT1 * KS + T1 * MT2 * KS) * 64;
        FBBEV_LAUNCH(k_history_weight_fragments, (nfrag + 255) / 256, 256, 0, stream, w1, w2, MT1, MT2, KS, T1, VM ? 1 : 0, w1f);
        if (C == 80)
            FBBEV_LAUNCH((k_history_conv_t<5, 5, ET, VM>), blocks, 256, lds, stream, feats, feats_stride_b,
                         (const float*)w1f, bias1, (const float*)w2f, bias2, T1, N, tiles_per_b, out);
        else
            FBBEV_LAUNCH((k_history_conv_t<1, 1, ET, VM>), blocks, 256, lds, stream, feats, feats_stride_b,
                         (const float*)w1f, bias1, (const float*)w2f, bias2, T1, N, tiles_per_b, out);
    } else if (VM)
        return FBBEV_E_UNSUPPORTED;
    else
        FBBEV_LAUNCH(k_history_conv<ET>, blocks, 256, lds, stream, feats, feats_stride_b, w1, bias1, w2, bias2,
                     T1, C, Cout, N, tiles_per_b, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_conv_e(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                    const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                    float* out, void* workspace, size_t workspace_bytes, int elem_type,
                                    fbbev_stream_t stream_) {
    if (B < 0 || T1 <= 0 || C <= 0 || Cout <= 0 || N < 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (B == 0 || N == 0) return 0;
    if (!feats || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (C % 16 != 0 || Cout % 16 != 0 || C > 16 * FBBEV_HC_MAX_TILES || Cout > 16 * FBBEV_HC_MAX_TILES)
        return FBBEV_E_UNSUPPORTED;
    if (feats_stride_b == 0) feats_stride_b = (long long)T1 * C * N;
    if (feats_stride_b < (long long)T1 * C * N) return FBBEV_E_BADARG;
    if (!aligned16(bias1)) return FBBEV_E_UNSUPPORTED;                                  // 16-byte bias loads
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    if (elem_type == 0) return history_conv_launch<0>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
    if (elem_type == 1) return history_conv_launch<1>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
    return history_conv_launch<2>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
}

// bf16-MFMA variant of the two fused convolutions (operands rounded to bf16, fp32 accumulate): C = Cout = 80 or 16
template <int ET, bool VM>
static int history_conv_bf16_launch(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                    const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                    float* out, void* workspace, size_t workspace_bytes, fbbev_rt_stream stream) {
    const int tiles_per_b = (N + 63) / 64;
    const long long blocks = (long long)B * tiles_per_b;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int MT1 = C / 16, MT2 = Cout / 16, KS = (C + 31) / 32;
    const size_t need = ((size_t)MT1 * KS + (size_t)T1 * MT2 * KS) * 64 * 8 * sizeof(unsigned short);
    if (!workspace || !aligned16(workspace) || workspace_bytes < need) return FBBEV_E_WORKSPACE;
    unsigned short* w1f = static_cast<unsigned short*>(workspace);
    unsigned short* w2f = w1f + (size_t)MT1 * KS * 64 * 8;
    const int nfrag = (MT1 * KS + T1 * MT2 * KS) * 64;
    FBBEV_LAUNCH(k_history_weight_fragments_bf16, (nfrag + 255) / 256, 256, 0, stream, w1, w2, MT1, MT2, C, T1, w1f);
    const size_t a2s = (size_t)((MT2 * KS * 64 + 255) / 256) * 256 * 8;                                 // padded staging buffer
    size_t lds = (2 * a2s + (size_t)4 * 16 * (KS * 32 + 8)) * sizeof(unsigned short);           // W2_t x 2 + Y rows
    static const size_t conv_pad = [] { const char* e = getenv("FBBEV_HISTORY_CONV_LDS_PAD_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();   // occupancy experiment knob
    if (conv_pad > lds) {
        lds = conv_pad;
        if (lds > 64 * 1024) fbbev_rt_allow_dyn_lds((const void*)k_history_conv_bf16<5, 5, ET, VM>, lds);
    }
    if (C == 80)
        FBBEV_LAUNCH((k_history_conv_bf16<5, 5, ET, VM>), blocks, 256, lds, stream, feats, feats_stride_b,
                     (const unsigned short*)w1f, bias1, (const unsigned short*)w2f, bias2, T1, N, tiles_per_b, out);
    else
        FBBEV_LAUNCH((k_history_conv_bf16<1, 1, ET, VM>), blocks, 256, lds, stream, feats, feats_stride_b,
                     (const unsigned short*)w1f, bias1, (const unsigned short*)w2f, bias2, T1, N, tiles_per_b, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_conv_bf16(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                       const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                       float* out, void* workspace, size_t workspace_bytes, int voxel_major, int elem_type,
                                       fbbev_stream_t stream_) {
    if (B < 0 || T1 <= 0 || C <= 0 || Cout <= 0 || N < 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (voxel_major != 0 && voxel_major != 1) return FBBEV_E_BADARG;
    if (B == 0 || N == 0) return 0;
    if (!feats || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (!((C == 80 && Cout == 80) || (C == 16 && Cout == 16))) return FBBEV_E_UNSUPPORTED;
    if (feats_stride_b == 0) feats_stride_b = (long long)T1 * C * N;
    if (feats_stride_b < (long long)T1 * C * N) return FBBEV_E_BADARG;
    if (voxel_major && (!aligned16(feats) || feats_stride_b % 8 != 0)) return FBBEV_E_UNSUPPORTED;   // 16-byte row pieces
    if (!aligned16(bias1)) return FBBEV_E_UNSUPPORTED;                                              // 16-byte bias loads
    if (voxel_major && (long long)N * C * (elem_type == 0 ? 4 : 2) >= (1ll << 32)) return FBBEV_E_UNSUPPORTED;   // 32-bit byte offsets in a frame
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
#define FBBEV_HCB(ET_, VM_) history_conv_bf16_launch<ET_, VM_>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream)
    if (voxel_major) return elem_type == 0 ? FBBEV_HCB(0, true) : elem_type == 1 ? FBBEV_HCB(1, true) : FBBEV_HCB(2, true);
    return elem_type == 0 ? FBBEV_HCB(0, false) : elem_type == 1 ? FBBEV_HCB(1, false) : FBBEV_HCB(2, false);
#undef FBBEV_HCB
}

// fp32-grade convolutions on the 16-bit MFMAs (split operands, history_conv_x3_kernels.h): voxel-major 16-bit ring
struct fbbev_conv_x3_plan { unsigned short* w1x; unsigned short* w2x; float* biasx; size_t pieces, part1; };

static int history_conv_x3_prepare(const void* feats, long long& feats_stride_b, const float* w1, const float* bias1, const float* w2,
                                   const float* bias2, int B, int T1, int C, int Cout, int N, float* out, void* workspace,
                                   size_t workspace_bytes, int elem_type, fbbev_rt_stream stream, fbbev_conv_x3_plan& pl) {
    if (B < 0 || T1 <= 0 || C <= 0 || Cout <= 0 || N < 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (B == 0 || N == 0) return 0;
    if (!feats || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (!((C == 80 && Cout == 80) || (C == 16 && Cout == 16)) || elem_type == 0) return FBBEV_E_UNSUPPORTED;
    if (feats_stride_b == 0) feats_stride_b = (long long)T1 * C * N;
    if (feats_stride_b < (long long)T1 * C * N) return FBBEV_E_BADARG;
    if (!aligned16(feats) || feats_stride_b % 8 != 0 || !aligned16(bias1)) return FBBEV_E_UNSUPPORTED;
    if ((long long)N * C * 2 >= (1ll << 32)) return FBBEV_E_UNSUPPORTED;                          // 32-bit byte offsets in a frame
    const int MT = C / 16, KS = (C + 31) / 32;
    const size_t part1 = (size_t)MT * KS * 64 * 8, part2 = part1;
    const size_t wbytes = (2 * part1 + (size_t)T1 * 2 * part2) * sizeof(unsigned short);
    const size_t need = wbytes + (size_t)B * T1 * C * sizeof(float);
    if (!workspace || !aligned16(workspace) || workspace_bytes < need) return FBBEV_E_WORKSPACE;
    pl.w1x = static_cast<unsigned short*>(workspace);
    pl.w2x = pl.w1x + 2 * part1;
    pl.biasx = reinterpret_cast<float*>(static_cast<char*>(workspace) + wbytes);
    const long long nprep = (long long)(MT * KS + T1 * MT * KS) * 64 + (long long)B * T1 * C;
    if (nprep >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    if (elem_type == 1)
        FBBEV_LAUNCH(k_history_weight_fragments_bf16x3<1>, (nprep + 255) / 256, 256, 0, stream, w1, w2, bias1, MT, MT, C, T1, B * T1, pl.w1x, pl.biasx);
    else
        FBBEV_LAUNCH(k_history_weight_fragments_bf16x3<2>, (nprep + 255) / 256, 256, 0, stream, w1, w2, bias1, MT, MT, C, T1, B * T1, pl.w1x, pl.biasx);
    FBBEV_CHECK_LAUNCH();
    pl.pieces = 2 * part2 / 8 + C / 4;
    pl.part1 = part1;
    return 0;
}

// n_seg segments of seg_len voxels, seg_stride apart from voxel seg0 on, in every sample (one segment of N voxels: the whole volume)
static int history_conv_x3_segments(const void* feats, long long feats_stride_b, const float* bias2, int B, int T1, int C, int N,
                                    float* out, int elem_type, const fbbev_conv_x3_plan& pl, int seg0, int seg_stride, int seg_len,
                                    int n_seg, int nw, fbbev_rt_stream stream) {
    if (seg_len <= 0 || n_seg <= 0) return 0;
    const int nt = 64 * nw;                                       // nw = 8: a workgroup fills a CU's registers; 4: half of them (the pipelined step)
    const int tile_voxels = 16 * nw * FBBEV_HX3_NV;
    const int tiles_per_seg = (seg_len + tile_voxels - 1) / tile_voxels;
    const long long tiles_per_b = (long long)tiles_per_seg * n_seg;
    const long long blocks = (long long)B * tiles_per_b;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    // one frame per workgroup barrier (FPB = 2, four W2 buffers: measured 2 % slower, profiles/r06_exp_history_step.md)
    const size_t lds = ((size_t)2 * ((pl.pieces + nt - 1) / nt) * nt * 8 + 2 * pl.part1) * sizeof(unsigned short);
#define FBBEV_HX3(MT_, ET_, PF_, NW_, FPB_)                                                                            \
    do {                                                                                                              \
        int e_ = fbbev_rt_allow_dyn_lds((const void*)k_history_conv_bf16x3<MT_, MT_, ET_, PF_, NW_, FPB_>, lds);        \
        if (e_) return e_;                                                                                            \
        FBBEV_LAUNCH((k_history_conv_bf16x3<MT_, MT_, ET_, PF_, NW_, FPB_>), blocks, 64 * NW_, lds, stream, feats, feats_stride_b, \
                     (const unsigned short*)pl.w1x, (const float*)pl.biasx, (const unsigned short*)pl.w2x, bias2, T1, N, \
                     (int)tiles_per_b, out, seg0, seg_stride, seg_len, tiles_per_seg);                                 \
    } while (0)
    /* two frames of X in flight: three spill (256 registers at 2 waves / SIMD) */
#define FBBEV_HX3P(MT_, ET_) do { if (nw == 4) FBBEV_HX3(MT_, ET_, 2, 4, 1); else FBBEV_HX3(MT_, ET_, 2, 8, 1); } while (0)
    if (C == 80) { if (elem_type == 1) FBBEV_HX3P(5, 1); else FBBEV_HX3P(5, 2); }
    else { if (elem_type == 1) FBBEV_HX3P(1, 1); else FBBEV_HX3P(1, 2); }
#undef FBBEV_HX3P
#undef FBBEV_HX3
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_history_conv_bf16x3(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                         const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                         float* out, void* workspace, size_t workspace_bytes, int elem_type,
                                         fbbev_stream_t stream_) {
    fbbev_conv_x3_plan pl{};
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    const int e = history_conv_x3_prepare(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace,
                                          workspace_bytes, elem_type, stream, pl);
    if (e || B <= 0 || N <= 0) return e;
    static const int nw = [] { const char* e = getenv("FBBEV_HX3_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();     // tuning: 256-thread workgroups
    return history_conv_x3_segments(feats, feats_stride_b, bias2, B, T1, C, N, out, elem_type, pl, 0, 0, N, 1, nw, stream);
}

// One history step on a 16-bit voxel-major ring as ONE kernel (history_fused_x3_kernels.h): every MFMA wave warps its own operands.
// next[:, 0] must hold the current frame; writes next[:, 1:] (== fbbev_history_warp_vm) and out (== fbbev_history_conv_bf16x3).
extern "C" int fbbev_history_fused_x3_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                                         const float* rt_flow, const float* w1, const float* bias1, const float* w2,
                                         const float* bias2, int B, int T, int C, int Cout, int Z, int Y, int X, float* out,
                                         void* workspace, size_t workspace_bytes, int elem_type, fbbev_stream_t stream_) {
    if (B < 0 || T <= 0 || C <= 0 || Cout <= 0 || Z <= 0 || Y <= 0 || X <= 0) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!history || !next || !rt_flow || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (C != 80 || Cout != 80 || (elem_type != 1 && elem_type != 2) || Z < 2 || Y < 2 || X < 2) return FBBEV_E_UNSUPPORTED;
    const long long N = (long long)Z * Y * X, frame = N * C;
    const int T1 = T + 1;
    if (history_stride_b == 0) history_stride_b = (long long)T * frame;
    if (next_stride_b == 0) next_stride_b = (long long)T1 * frame;
    if (history_stride_b < (long long)T * frame || next_stride_b < (long long)T1 * frame) return FBBEV_E_BADARG;
    if (history_stride_b % 8 != 0 || next_stride_b % 8 != 0 || !aligned16(history) || !aligned16(next)) return FBBEV_E_UNSUPPORTED;
    if (frame * 2 >= (1ll << 32) || N >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;              // 32-bit byte offsets inside a frame
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    fbbev_conv_x3_plan cp{};
    long long fs = next_stride_b;
    if (workspace_bytes < 64) return FBBEV_E_WORKSPACE;
    int e = history_conv_x3_prepare(next, fs, w1, bias1, w2, bias2, B, T1, C, Cout, (int)N, out, workspace, workspace_bytes - 64, elem_type, stream, cp);
    if (e) return e;
    void* dump = static_cast<char*>(workspace) + ((workspace_bytes - 64) & ~(size_t)15);       // 16 bytes nobody reads (the tail of the workspace)
    const int n_xt = (X + 15) / 16, n_yt = (Y + 7) / 8;           // bricks of 16 x 8 voxels of one z plane, z fastest
    const long long blocks = (long long)B * n_yt * n_xt * Z;
    if (blocks >= (1ll << 31) - 8) return FBBEV_E_UNSUPPORTED;
    const int per_xcd = (int)((blocks + 7) / 8);
    // two W2 / bias blocks + W1 (hi | lo) + two operand tiles [128 voxels][FBBEV_HFX_PITCH] + the voxel table [128][16 dwords]: 154 KB
    const size_t lds = ((size_t)2 * ((cp.pieces + 511) / 512) * 512 * 8 + 2 * cp.part1 + (size_t)2 * 128 * FBBEV_HFX_PITCH) * sizeof(unsigned short) +
                       (size_t)128 * 16 * sizeof(unsigned int);
#define FBBEV_HFX(ET_)                                                                                                 \
    do {                                                                                                              \
        int e_ = fbbev_rt_allow_dyn_lds((const void*)k_history_fused_x3<ET_>, lds);                                    \
        if (e_) return e_;                                                                                            \
        FBBEV_LAUNCH((k_history_fused_x3<ET_>), (long long)per_xcd * 8, 512, lds, stream, history, history_stride_b,   \
                     next, next_stride_b, rt_flow, (const unsigned short*)cp.w1x, (const float*)cp.biasx,             \
                     (const unsigned short*)cp.w2x, bias2, T1, Z, Y, X, n_xt, n_yt, per_xcd, (int)blocks, out, dump);  \
    } while (0)
    if (elem_type == 1) FBBEV_HFX(1); else FBBEV_HFX(2);
#undef FBBEV_HFX
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// One history step on a 16-bit voxel-major ring as a two-stream pipeline: the warp of a band of grid rows into next[:, 1:] on the
// caller's stream, the split-operand convolutions of the band before it on a second stream -- a bandwidth-bound gather kernel and
// an MFMA-bound one that leave each other's resource idle when they run back to back.  next[:, 0] must hold the current frame.
// The same kernels, operands and results as fbbev_history_warp_vm + fbbev_history_conv_bf16x3, bit for bit.
struct fbbev_side_stream { fbbev_rt_stream stream; fbbev_rt_event ev[2 + 64]; bool ok; };

static fbbev_side_stream* history_side_stream() {
    static fbbev_side_stream table[16];
    static bool made[16];
    const int d = fbbev_rt_device();
    if (d < 0 || d >= 16) return nullptr;
    static std::mutex mu;                                    // (first use from two host threads; the stream and its events are per DEVICE:
    std::lock_guard<std::mutex> lock(mu);                    //  callers serialise their calls per device, as with any stream-ordered workspace)
    if (!made[d]) {
        made[d] = true;
        // the convolutions' stream is created at the device's highest priority: built to let their workgroups in ahead of the thousands
        // of queued warp workgroups -- measured: no effect on the dispatch (profiles/r06_exp_history_step.md).
        // FBBEV_HISTORY_STEP_PRIORITY=0: equal priorities
        const char* pe = getenv("FBBEV_HISTORY_STEP_PRIORITY");
        table[d].ok = fbbev_rt_stream_create(&table[d].stream, pe && atoi(pe) == 0 ? 0 : 1) == 0;
        for (int i = 0; i < 2 + 64 && table[d].ok; ++i) table[d].ok = fbbev_rt_event_create(&table[d].ev[i]) == 0;
    }
    return table[d].ok ? &table[d] : nullptr;
}

extern "C" int fbbev_history_step_x3_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                                        const float* rt_flow, const float* w1, const float* bias1, const float* w2,
                                        const float* bias2, int B, int T, int C, int Cout, int Z, int Y, int X, float* out,
                                        void* workspace, size_t workspace_bytes, int elem_type, int chunks,
                                        fbbev_stream_t stream_) {
    if (B < 0 || T <= 0 || C <= 0 || Cout <= 0 || Z <= 0 || Y <= 0 || X <= 0 || chunks < 0 || chunks > 64) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!history || !next || !rt_flow || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (elem_type != 1 && elem_type != 2) return FBBEV_E_UNSUPPORTED;
    const long long N = (long long)Z * Y * X, frame = N * C;
    if (N >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int T1 = T + 1;
    if (next_stride_b == 0) next_stride_b = (long long)T1 * frame;
    if (next_stride_b < (long long)T1 * frame) return FBBEV_E_BADARG;
    const int esz = 2;
    void* warped = static_cast<char*>(next) + frame * esz;                                       // next[:, 1:]
    fbbev_warp_vm_plan wp{};
    long long hs = history_stride_b, ns = next_stride_b;
    int e = history_warp_vm_plan(history, hs, rt_flow, B, T, C, Z, Y, X, warped, ns, elem_type, wp);
    if (e) return e;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    fbbev_conv_x3_plan cp{};
    long long fs = next_stride_b;
    e = history_conv_x3_prepare(next, fs, w1, bias1, w2, bias2, B, T1, C, Cout, (int)N, out, workspace, workspace_bytes, elem_type, stream, cp);
    if (e) return e;
    if (chunks == 0) chunks = 10;
    // the convolutions of the pipeline in 256-thread workgroups: half a CU's registers, so that warp waves share the CU with them
    static const int step_nw = [] { const char* e = getenv("FBBEV_HISTORY_STEP_WAVES"); return e && atoi(e) == 8 ? 8 : 4; }();
    if (chunks > wp.nyb) chunks = wp.nyb;
    fbbev_side_stream* side = chunks > 1 ? history_side_stream() : nullptr;
    if (!side) {                                                                                  // one chunk: the two kernels back to back
        e = history_warp_vm_bands(history, hs, rt_flow, B, T, C, Z, Y, X, warped, ns, elem_type, wp, 0, wp.nyb, stream);
        if (e) return e;
        return history_conv_x3_segments(next, fs, bias2, B, T1, C, (int)N, out, elem_type, cp, 0, 0, (int)N, 1, 8, stream);
    }
#define FBBEV_RT(x) do { const int e_ = (x); if (e_) return e_; } while (0)
    FBBEV_RT(fbbev_rt_event_record(side->ev[0], stream));
    FBBEV_RT(fbbev_rt_stream_wait(side->stream, side->ev[0]));
    for (int i = 0; i < chunks; ++i) {
        const int yb0 = (int)((long long)wp.nyb * i / chunks), yb1 = (int)((long long)wp.nyb * (i + 1) / chunks);
        e = history_warp_vm_bands(history, hs, rt_flow, B, T, C, Z, Y, X, warped, ns, elem_type, wp, yb0, yb1 - yb0, stream);
        if (e) return e;
        FBBEV_RT(fbbev_rt_event_record(side->ev[2 + i], stream));
        FBBEV_RT(fbbev_rt_stream_wait(side->stream, side->ev[2 + i]));
        const int y0 = yb0 * wp.YB, y1 = yb1 * wp.YB < Y ? yb1 * wp.YB : Y;
        e = history_conv_x3_segments(next, fs, bias2, B, T1, C, (int)N, out, elem_type, cp, y0 * X, Y * X, (y1 - y0) * X, Z, step_nw, side->stream);
        if (e) return e;
    }
    FBBEV_RT(fbbev_rt_event_record(side->ev[1], side->stream));
    FBBEV_RT(fbbev_rt_stream_wait(stream, side->ev[1]));
#undef FBBEV_RT
    return 0;
}

// ------------------------------------------------------------------------------ row-wise linear layers, split-operand bf16 MFMA
extern "C" size_t fbbev_rows_linear_x3_fragment_bytes(int in_features, int out_features) {
    if (in_features <= 0 || out_features <= 0) return 0;
    const size_t n_oc = (size_t)(out_features + 127) / 128, n_kc = (size_t)(in_features + 127) / 128;
    return n_oc * n_kc * 8 * FBBEV_RL_TILE_ELEMS * sizeof(unsigned short);
}

extern "C" int fbbev_rows_linear_x3_fragments(const float* weight, int in_features, int out_features, void* fragments,
                                              size_t fragment_bytes, fbbev_stream_t stream_) {
    if (in_features <= 0 || out_features <= 0 || !weight || !fragments) return FBBEV_E_BADARG;
    if (!aligned16(fragments) || fragment_bytes < fbbev_rows_linear_x3_fragment_bytes(in_features, out_features)) return FBBEV_E_WORKSPACE;
    const int n_oc = (out_features + 127) / 128, n_kc = (in_features + 127) / 128;
    const long long n = (long long)n_oc * n_kc * 8 * 4 * 64;
    FBBEV_LAUNCH(k_rows_linear_x3_fragments, (n + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, weight, out_features, in_features,
                 n_oc, n_kc, static_cast<unsigned short*>(fragments));
    FBBEV_CHECK_LAUNCH();
    return 0;
}

static bool rows_linear_nt1() { static const bool v = [] { const char* e = getenv("FBBEV_ROWS_LINEAR_NT"); return e && atoi(e) == 1; }(); return v; }   // tuning knob, read once
static int rows_linear_x3_impl(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                               int in_features, int out_features, int relu, float* out, long long out_row_stride,
                               const float* addend, long long addend_row_stride, long long addend_period, fbbev_stream_t stream_,
                               int plane_S = 0, int plane_TS = 0, const float* res = nullptr, long long ld_res = 0,
                               const float* ln_w = nullptr, const float* ln_b = nullptr, float ln_eps = 0.f,
                               const float* mask = nullptr, long long ld_mask = 0, bool train_epi = false);

extern "C" int fbbev_rows_linear_x3(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                                    int in_features, int out_features, int relu, float* out, long long out_row_stride,
                                    fbbev_stream_t stream_) {
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, relu, out, out_row_stride, nullptr,
                               0, 1, stream_);
}

extern "C" int fbbev_rows_linear_x3_add(const float* x, long long x_row_stride, const float* addend, long long addend_row_stride,
                                        long long addend_period, const void* fragments, const float* bias, long long rows,
                                        int in_features, int out_features, int relu, float* out, long long out_row_stride,
                                        fbbev_stream_t stream_) {
    if (!addend || addend_period <= 0) return FBBEV_E_BADARG;
    if (addend_row_stride == 0) addend_row_stride = in_features;
    if (addend_row_stride < in_features) return FBBEV_E_BADARG;
    if (addend_row_stride % 4 != 0 || !aligned16(addend)) return FBBEV_E_UNSUPPORTED;
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, relu, out, out_row_stride, addend,
                               addend_row_stride, addend_period, stream_);
}

static int rows_linear_x3_impl(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                               int in_features, int out_features, int relu, float* out, long long out_row_stride,
                               const float* addend, long long addend_row_stride, long long addend_period, fbbev_stream_t stream_,
                               int plane_S, int plane_TS, const float* res, long long ld_res, const float* ln_w, const float* ln_b,
                               float ln_eps, const float* mask, long long ld_mask, bool train_epi) {
    if (rows < 0 || in_features <= 0 || out_features <= 0) return FBBEV_E_BADARG;
    if (rows == 0) return 0;
    if (!x || !fragments || !out) return FBBEV_E_BADARG;
    if (x_row_stride == 0) x_row_stride = in_features;
    if (out_row_stride == 0) out_row_stride = out_features;
    if (x_row_stride < in_features || out_row_stride < out_features) return FBBEV_E_BADARG;
    if (in_features % 8 != 0 || out_features % 4 != 0 || x_row_stride % 4 != 0 || out_row_stride % 4 != 0 || !aligned16(x) ||
        !aligned16(out) || !aligned16(fragments) || (bias && !aligned16(bias))) return FBBEV_E_UNSUPPORTED;
    const int n_oc = (out_features + 127) / 128, n_kc = (in_features + 127) / 128;
    const long long tiles = (rows + 127) / 128;
    if (tiles * n_oc >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int nmt = out_features >= 128 ? 8 : (out_features + 15) / 16;
    const size_t lds = (size_t)nmt * FBBEV_RL_TILE_ELEMS * sizeof(unsigned short);
    int e = ln_w ? fbbev_rt_allow_dyn_lds((const void*)k_rows_linear_x3<2, true>, lds) : fbbev_rt_allow_dyn_lds((const void*)k_rows_linear_x3<2, false>, lds);
    if (e) return e;
    // consecutive row tiles per workgroup (fragments staged once) as long as ~4 workgroups per CU remain
    long long RT = n_kc == 1 ? tiles * n_oc / 1024 : 1;
    RT = RT < 1 ? 1 : (RT > 8 ? 8 : RT);
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: lets a small case walk several row tiles per workgroup
    { const char* e_rt = getenv("FBBEV_ROWS_LINEAR_RT"); if (e_rt && n_kc == 1 && atoi(e_rt) >= 1 && atoi(e_rt) <= 8) RT = atoi(e_rt); }
#endif
    const long long groups = (tiles + RT - 1) / RT;
    // round 6: the persistent form (k_rows_linear_x3p: fragments staged once per workgroup, the next tile's rows in flight under the
    // MFMAs, fragment reads under the MFMAs of the tile before) for one K chunk, no addend, no LayerNorm tail.  Same bits.
    // FBBEV_ROWS_LINEAR_P=0: off (A/B knob)
#ifdef FBBEV_TEST_OVERRIDES
    const bool persist = [] { const char* e_ = getenv("FBBEV_ROWS_LINEAR_P"); return !(e_ && atoi(e_) == 0); }();
#else
    static const bool persist = [] { const char* e_ = getenv("FBBEV_ROWS_LINEAR_P"); return !(e_ && atoi(e_) == 0); }();
#endif
    // (the training epilogue at 8 output tiles x 4 k-steps -- 96 < I <= 128 and O > 80: no layer of the model -- would spill: old kernel)
    if (persist && n_kc == 1 && !ln_w && !addend && !(rows_linear_nt1() && !train_epi) &&
        !(train_epi && out_features > 80 && in_features > 96)) {
        const int nmtp = out_features <= 80 ? 5 : 8, ksp = in_features <= 96 ? 3 : 4;
        long long n_slots = 512 / n_oc;
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: lets a small case walk several row tiles per workgroup
        { const char* e_sl = getenv("FBBEV_ROWS_LINEAR_SLOTS"); if (e_sl && atoi(e_sl) >= 1) n_slots = atoi(e_sl); }
#endif
        if (n_slots < 1) n_slots = 1;
        if (n_slots > tiles) n_slots = tiles;
        const size_t ldsp = (size_t)nmtp * FBBEV_RL_TILE_ELEMS * sizeof(unsigned short) + (size_t)16 * nmtp * sizeof(float);
#define FBBEV_RLP(NMT_, KS_, EPI_)                                                                                      \
    do {                                                                                                              \
        e = fbbev_rt_allow_dyn_lds((const void*)k_rows_linear_x3p<NMT_, KS_, EPI_>, ldsp);                              \
        if (e) return e;                                                                                              \
        FBBEV_LAUNCH((k_rows_linear_x3p<NMT_, KS_, EPI_>), n_slots * n_oc, 256, ldsp, (fbbev_rt_stream)stream_, x, x_row_stride, \
                     static_cast<const unsigned short*>(fragments), bias, out, out_row_stride, rows, in_features, out_features, relu, \
                     n_oc, (int)n_slots, plane_S, plane_TS, res, ld_res, mask, ld_mask);                              \
    } while (0)
        if (train_epi) {
            if (nmtp == 5) { if (ksp == 3) FBBEV_RLP(5, 3, 1); else FBBEV_RLP(5, 4, 1); }
            else FBBEV_RLP(8, 3, 1);
        } else {
            if (nmtp == 5) { if (ksp == 3) FBBEV_RLP(5, 3, 0); else FBBEV_RLP(5, 4, 0); }
            else { if (ksp == 3) FBBEV_RLP(8, 3, 0); else FBBEV_RLP(8, 4, 0); }
        }
#undef FBBEV_RLP
        FBBEV_CHECK_LAUNCH();
        return 0;
    }
    if (train_epi) {
        e = fbbev_rt_allow_dyn_lds((const void*)k_rows_linear_x3<2, false, 1>, lds);
        if (e) return e;
        FBBEV_LAUNCH((k_rows_linear_x3<2, false, 1>), groups * n_oc, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,
                     static_cast<const unsigned short*>(fragments), bias, out, out_row_stride, rows, in_features, out_features, relu,
                     n_kc, n_oc, (int)RT, addend, addend_row_stride, addend_period, 0, 0, res, ld_res, (const float*)nullptr,
                     (const float*)nullptr, 0.f, mask, ld_mask);
    } else if (ln_w) {
        if (n_oc != 1) return FBBEV_E_UNSUPPORTED;
        FBBEV_LAUNCH((k_rows_linear_x3<2, true>), groups * n_oc, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,
                     static_cast<const unsigned short*>(fragments), bias, out, out_row_stride, rows, in_features, out_features, relu,
                     n_kc, n_oc, (int)RT, addend, addend_row_stride, addend_period, plane_S, plane_TS, res, ld_res, ln_w, ln_b, ln_eps,
                     (const float*)nullptr, 0ll);
    } else if (rows_linear_nt1() && n_kc == 1) {
        // 16 rows per wave (64 per workgroup): half the accumulators / row pieces in registers -> more waves per SIMD
        const long long tiles1 = (rows + 63) / 64;
        long long RT1 = tiles1 * n_oc / 2048;
        RT1 = RT1 < 1 ? 1 : (RT1 > 8 ? 8 : RT1);
        e = fbbev_rt_allow_dyn_lds((const void*)k_rows_linear_x3<1, false>, lds);
        if (e) return e;
        FBBEV_LAUNCH((k_rows_linear_x3<1, false>), (tiles1 + RT1 - 1) / RT1 * n_oc, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,
                     static_cast<const unsigned short*>(fragments), bias, out, out_row_stride, rows, in_features, out_features, relu,
                     n_kc, n_oc, (int)RT1, addend, addend_row_stride, addend_period, plane_S, plane_TS, res, ld_res, ln_w, ln_b, ln_eps,
                     (const float*)nullptr, 0ll);
    } else {
        FBBEV_LAUNCH((k_rows_linear_x3<2, false>), groups * n_oc, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,
                     static_cast<const unsigned short*>(fragments), bias, out, out_row_stride, rows, in_features, out_features, relu,
                     n_kc, n_oc, (int)RT, addend, addend_row_stride, addend_period, plane_S, plane_TS, res, ld_res, ln_w, ln_b, ln_eps,
                     (const float*)nullptr, 0ll);
    }
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// training epilogue (round 6): out = ((x W^T + b) [ReLU]) * [mask > 0] + residual -- the forward recomputations and dgrads of the
// encoder layer's backward with their neighbouring element-wise passes (residual adds, ReLU's threshold_backward, gradient sums)
extern "C" int fbbev_rows_linear_x3_train(const float* x, long long x_row_stride, const float* addend, long long addend_row_stride,
                                          long long addend_period, const void* fragments, const float* bias, long long rows,
                                          int in_features, int out_features, int relu, const float* residual, long long residual_row_stride,
                                          const float* mask, long long mask_row_stride, float* out, long long out_row_stride,
                                          fbbev_stream_t stream_) {
    if (out_features <= 0 || in_features <= 0) return FBBEV_E_BADARG;
    if (addend) {
        if (addend_period <= 0) return FBBEV_E_BADARG;
        if (addend_row_stride == 0) addend_row_stride = in_features;
        if (addend_row_stride < in_features) return FBBEV_E_BADARG;
        if (addend_row_stride % 4 != 0 || !aligned16(addend)) return FBBEV_E_UNSUPPORTED;
    } else {
        addend_period = 1;
    }
    if (residual) {
        if (residual_row_stride == 0) residual_row_stride = out_features;
        if (residual_row_stride < out_features) return FBBEV_E_BADARG;
        if (residual_row_stride % 4 != 0 || !aligned16(residual)) return FBBEV_E_UNSUPPORTED;
    }
    if (mask) {
        if (mask_row_stride == 0) mask_row_stride = out_features;
        if (mask_row_stride < out_features) return FBBEV_E_BADARG;
        if (mask_row_stride % 4 != 0 || !aligned16(mask)) return FBBEV_E_UNSUPPORTED;
    }
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, relu, out, out_row_stride, addend,
                               addend_row_stride, addend_period, stream_, 0, 0, residual, residual_row_stride, nullptr, nullptr, 0.f,
                               mask, mask_row_stride, true);
}

// out = LayerNorm(x W^T + b [+ residual]) over the out_features outputs of a row, weight / bias / eps of torch.nn.LayerNorm: the
// `output_proj -> + residual -> norm` tail of the encoder layer's attention blocks and of its FFN (bevformer_encoder.py:250-377
// with operation_order (attn, norm, ...)) as ONE kernel -- the LayerNorm rides in the GEMM's store epilogue.  out_features <= 128
// (a workgroup must hold whole output rows).
extern "C" int fbbev_rows_linear_x3_ln(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                                       int in_features, int out_features, const float* residual, long long residual_row_stride,
                                       const float* ln_weight, const float* ln_bias, float ln_eps, float* out,
                                       long long out_row_stride, fbbev_stream_t stream_) {
    if (!ln_weight || !ln_bias || out_features <= 0) return FBBEV_E_BADARG;
    if (out_features > 128) return FBBEV_E_UNSUPPORTED;
    if (residual) {
        if (residual_row_stride == 0) residual_row_stride = out_features;
        if (residual_row_stride < out_features) return FBBEV_E_BADARG;
        if (residual_row_stride % 4 != 0 || !aligned16(residual)) return FBBEV_E_UNSUPPORTED;
    }
    if (!aligned16(ln_weight) || !aligned16(ln_bias)) return FBBEV_E_UNSUPPORTED;
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, 0, out, out_row_stride, nullptr, 0, 1,
                               stream_, 0, 0, residual, residual_row_stride, ln_weight, ln_bias, ln_eps);
}

// y = x W^T + b written as HEAD PLANES: rows = (B*Ncam) x S tokens, out_features = M * head_dim (module order (head, channel));
// out (B*Ncam, M, S, head_dim) -- the camera-token layout of fbbev_da_cross_attn_fused (value_proj of the cross-attention).
extern "C" int fbbev_rows_linear_x3_planes(const float* x, long long x_row_stride, const void* fragments, const float* bias,
                                           long long rows, int in_features, int out_features, int tokens_per_image, int head_dim,
                                           float* out, fbbev_stream_t stream_) {
    if (tokens_per_image <= 0 || head_dim <= 0 || out_features <= 0) return FBBEV_E_BADARG;
    // (the head-plane epilogue divides by head_dim through a 32-bit reciprocal, exact for outputs < 2^16, and packs the element
    // type above bit 16 of its head-width argument)
    if (head_dim % 2 != 0 || out_features % head_dim != 0 || rows % tokens_per_image != 0 || head_dim > 0xffff || out_features > 0xffff)
        return FBBEV_E_UNSUPPORTED;
    if (out && ((uintptr_t)out & 7) != 0) return FBBEV_E_UNSUPPORTED;
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, 0, out, 0, nullptr, 0, 1, stream_,
                               tokens_per_image, head_dim);
}

// The same projection with the planes STORED in 16 bits (elem_type 1 bf16, 2 fp16; 0 = fbbev_rows_linear_x3_planes): the fp32 result
// rounded once -- the camera-token storage option of the cross-attention (DA_SpatialCrossAttention.value_dtype) on head planes.
extern "C" int fbbev_rows_linear_x3_planes_e(const float* x, long long x_row_stride, const void* fragments, const float* bias,
                                             long long rows, int in_features, int out_features, int tokens_per_image, int head_dim,
                                             int elem_type, void* out, fbbev_stream_t stream_) {
    if (tokens_per_image <= 0 || head_dim <= 0 || out_features <= 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (head_dim % 2 != 0 || out_features % head_dim != 0 || rows % tokens_per_image != 0 || head_dim > 0xffff || out_features > 0xffff)
        return FBBEV_E_UNSUPPORTED;
    if (out && ((uintptr_t)out & (elem_type ? 3 : 7)) != 0) return FBBEV_E_UNSUPPORTED;
    if (out && !aligned16(out)) return FBBEV_E_UNSUPPORTED;
    return rows_linear_x3_impl(x, x_row_stride, fragments, bias, rows, in_features, out_features, 0, static_cast<float*>(out), 0, nullptr, 0,
                               1, stream_, tokens_per_image, head_dim | (elem_type << 16));
}

// The FFN pair of the encoder layer in one kernel: out = [LayerNorm](W2 relu(W1 x + b1) + b2 [+ residual]) -- mmcv FFN
// (bevformer_encoder.py:250-377 with ffn_cfgs: Linear + ReLU, Linear, add_identity) and, when ln_weight is given, the layer's
// following LayerNorm.  w1_fragments / w2_fragments = fbbev_rows_linear_x3_fragments of W1 (hidden, in) / W2 (out, hidden).
// Supported: in_features <= 96, hidden % 64 == 0, out_features <= 80 (the FB-OCC shape 80 -> 320 -> 80); FBBEV_E_UNSUPPORTED otherwise.
extern "C" int fbbev_rows_ffn_x3(const float* x, long long x_row_stride, const void* w1_fragments, const float* b1,
                                 const void* w2_fragments, const float* b2, long long rows, int in_features, int hidden,
                                 int out_features, const float* residual, long long residual_row_stride, const float* ln_weight,
                                 const float* ln_bias, float ln_eps, float* out, long long out_row_stride, fbbev_stream_t stream_) {
    if (rows < 0 || in_features <= 0 || hidden <= 0 || out_features <= 0) return FBBEV_E_BADARG;
    if (rows == 0) return 0;
    if (!x || !w1_fragments || !b1 || !w2_fragments || !b2 || !out || (ln_weight && !ln_bias)) return FBBEV_E_BADARG;
    if (x_row_stride == 0) x_row_stride = in_features;
    if (out_row_stride == 0) out_row_stride = out_features;
    if (x_row_stride < in_features || out_row_stride < out_features) return FBBEV_E_BADARG;
    if (residual) {
        if (residual_row_stride == 0) residual_row_stride = out_features;
        if (residual_row_stride < out_features) return FBBEV_E_BADARG;
    }
    if (in_features > 96 || in_features % 8 != 0 || hidden % FBBEV_FFN_HC != 0 || out_features > 80 || out_features % 4 != 0 ||
        x_row_stride % 4 != 0 || out_row_stride % 4 != 0 || !aligned16(x) || !aligned16(out) || !aligned16(w1_fragments) ||
        !aligned16(w2_fragments) || !aligned16(b1) || !aligned16(b2) || (residual && (residual_row_stride % 4 != 0 || !aligned16(residual))) ||
        (ln_weight && (!aligned16(ln_weight) || !aligned16(ln_bias)))) return FBBEV_E_UNSUPPORTED;
    const long long wgs = (rows + 127) / 128;
    if (wgs >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    // hidden units per chunk: 32 (three workgroups per CU) measured 1.480 vs 1.508 ms for 64 in the S3 scope (profiles/r04_time_fb_ffn_hc.jsonl)
#ifdef FBBEV_TEST_OVERRIDES   // CPU emulator build: the tests switch the chunk inside one process
    const int hc = [] { const char* e = getenv("FBBEV_FFN_HC"); return e ? atoi(e) : 32; }();
#else
    static const int hc = [] { const char* e = getenv("FBBEV_FFN_HC"); return e ? atoi(e) : 32; }();   // tuning knob, read once
#endif
    const int n_kc2 = (hidden + 127) / 128;
#define FBBEV_FFN(LN_, HC_)                                                                                              \
    do {                                                                                                               \
        const size_t lds = (size_t)fbbev_ffn_lds_bytes<3, 5, HC_>() + (size_t)hidden * 4; /* + b1 */                                                 \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_rows_ffn_x3<3, 5, LN_, HC_>, lds);                               \
        if (e) return e;                                                                                               \
        FBBEV_LAUNCH((k_rows_ffn_x3<3, 5, LN_, HC_>), wgs, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,          \
                     static_cast<const unsigned short*>(w1_fragments), b1, static_cast<const unsigned short*>(w2_fragments), b2,  \
                     out, out_row_stride, rows, in_features, hidden, out_features, n_kc2, residual, residual_row_stride,     \
                     ln_weight, ln_bias, ln_eps, fbbev_ffn_pre{});                                                     \
    } while (0)
    if (hc == 64) { if (ln_weight) FBBEV_FFN(true, 64); else FBBEV_FFN(false, 64); }
    else { if (ln_weight) FBBEV_FFN(true, 32); else FBBEV_FFN(false, 32); }
#undef FBBEV_FFN
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// The cross-attention block's tail AND the FFN block of the encoder layer in one kernel (round 5):
//   y1  = LayerNorm0(x W0^T + b0 + residual0)                 output_proj + residual + norm   (bevformer_encoder.py:250-377, ops 'cross_attn', 'norm')
//   out = LayerNorm1(y1 + W2 relu(W1 y1 + b1) + b2)           mmcv FFN (add_identity) + norm  (ops 'ffn', 'norm')
// x (rows, E) = the attention slots, residual0 (rows, E) = the block's input rows.  Supported: E in {16, 32, 48, 64, 80}, hidden % 64 == 0.
static int rows_tail_ffn_x3_impl(const float* x, long long x_row_stride, const void* w0_fragments, const float* b0,
                                 const float* residual0, long long residual0_row_stride, const float* ln0_weight,
                                 const float* ln0_bias, float ln0_eps, const void* w1_fragments, const float* b1,
                                 const void* w2_fragments, const float* b2, long long rows, int embed, int hidden,
                                 const float* ln1_weight, const float* ln1_bias, float ln1_eps, float* out,
                                 long long out_row_stride, long long tokens_per_image, fbbev_stream_t stream_) {
    if (rows < 0 || embed <= 0 || hidden <= 0 || tokens_per_image < 0) return FBBEV_E_BADARG;
    if (rows == 0) return 0;
    if (tokens_per_image > 0) {                            // planes: out (rows / S, embed, S); the kernel takes S as a negative row stride
        if (rows % tokens_per_image != 0) return FBBEV_E_BADARG;
        if (rows >= (1ll << 31) || tokens_per_image >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;      // 32-bit row arithmetic in the store
        out_row_stride = embed;                            // (passes the checks below)
    }
    if (!x || !w0_fragments || !b0 || !ln0_weight || !ln0_bias || !w1_fragments || !b1 || !w2_fragments || !b2 || !ln1_weight ||
        !ln1_bias || !out) return FBBEV_E_BADARG;
    if (x_row_stride == 0) x_row_stride = embed;
    if (out_row_stride == 0) out_row_stride = embed;
    if (residual0 && residual0_row_stride == 0) residual0_row_stride = embed;
    if (x_row_stride < embed || out_row_stride < embed || (residual0 && residual0_row_stride < embed)) return FBBEV_E_BADARG;
    if (embed > 80 || embed % 16 != 0 || hidden % FBBEV_FFN_HC != 0 || x_row_stride % 4 != 0 || out_row_stride % 4 != 0 ||
        !aligned16(x) || !aligned16(out) || !aligned16(w0_fragments) || !aligned16(w1_fragments) || !aligned16(w2_fragments) ||
        !aligned16(b0) || !aligned16(b1) || !aligned16(b2) || !aligned16(ln0_weight) || !aligned16(ln0_bias) || !aligned16(ln1_weight) ||
        !aligned16(ln1_bias) || (residual0 && (residual0_row_stride % 4 != 0 || !aligned16(residual0)))) return FBBEV_E_UNSUPPORTED;
    const long long wgs = (rows + 127) / 128;
    if (wgs >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const int n_kc2 = (hidden + 127) / 128;
    fbbev_ffn_pre pre;
    pre.w0f = static_cast<const unsigned short*>(w0_fragments); pre.b0 = b0; pre.res0 = residual0; pre.ld_res0 = residual0_row_stride;
    pre.ln0_w = ln0_weight; pre.ln0_b = ln0_bias; pre.eps0 = ln0_eps;
#define FBBEV_TFFN(HC_)                                                                                                \
    do {                                                                                                              \
        const size_t lds = (size_t)fbbev_ffn_pre_lds_bytes<3, 5, HC_>() + (size_t)hidden * 4; /* + b1 */                                             \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_rows_ffn_x3<3, 5, true, HC_, true>, lds);                        \
        if (e) return e;                                                                                              \
        FBBEV_LAUNCH((k_rows_ffn_x3<3, 5, true, HC_, true>), wgs, 256, lds, (fbbev_rt_stream)stream_, x, x_row_stride,   \
                     static_cast<const unsigned short*>(w1_fragments), b1, static_cast<const unsigned short*>(w2_fragments), b2, \
                     out, ldo, rows, embed, hidden, embed, n_kc2, (const float*)nullptr, (long long)0, ln1_weight,             \
                     ln1_bias, ln1_eps, pre);                                                                         \
    } while (0)
    const long long ldo = tokens_per_image > 0 ? -tokens_per_image : out_row_stride;
    FBBEV_TFFN(32);      // (hidden chunks of 64 -- FBBEV_TAIL_FFN_HC=64 until round 5 -- measured no gain: 1.392 vs 1.392 ms S3; gone)
#undef FBBEV_TFFN
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_rows_tail_ffn_x3(const float* x, long long x_row_stride, const void* w0_fragments, const float* b0,
                                      const float* residual0, long long residual0_row_stride, const float* ln0_weight,
                                      const float* ln0_bias, float ln0_eps, const void* w1_fragments, const float* b1,
                                      const void* w2_fragments, const float* b2, long long rows, int embed, int hidden,
                                      const float* ln1_weight, const float* ln1_bias, float ln1_eps, float* out,
                                      long long out_row_stride, fbbev_stream_t stream_) {
    return rows_tail_ffn_x3_impl(x, x_row_stride, w0_fragments, b0, residual0, residual0_row_stride, ln0_weight, ln0_bias, ln0_eps,
                                 w1_fragments, b1, w2_fragments, b2, rows, embed, hidden, ln1_weight, ln1_bias, ln1_eps, out,
                                 out_row_stride, 0, stream_);
}

// The same with the result written as planes: out (rows / tokens_per_image, embed, tokens_per_image) -- the (B, C, Y, X) refined BEV
// the final pooling re-adds (backward_projection.py:129: permute + view + contiguous), without the transposing pass behind the layer.
extern "C" int fbbev_rows_tail_ffn_x3_planes(const float* x, long long x_row_stride, const void* w0_fragments, const float* b0,
                                             const float* residual0, long long residual0_row_stride, const float* ln0_weight,
                                             const float* ln0_bias, float ln0_eps, const void* w1_fragments, const float* b1,
                                             const void* w2_fragments, const float* b2, long long rows, int embed, int hidden,
                                             const float* ln1_weight, const float* ln1_bias, float ln1_eps,
                                             long long tokens_per_image, float* out, fbbev_stream_t stream_) {
    if (tokens_per_image <= 0) return FBBEV_E_BADARG;
    return rows_tail_ffn_x3_impl(x, x_row_stride, w0_fragments, b0, residual0, residual0_row_stride, ln0_weight, ln0_bias, ln0_eps,
                                 w1_fragments, b1, w2_fragments, b2, rows, embed, hidden, ln1_weight, ln1_bias, ln1_eps, out, 0,
                                 tokens_per_image, stream_);
}

// Warp + new ring + both convolutions in ONE kernel (k_history_fused_bf16): history (B,T,N,C) -> next ring slots 1..T of
// `next` (B,T+1,N,C) (slot 0 = the current frame, stored by the caller with fbbev_history_frame_vm BEFORE this call) and
// out (B,Cout,N) = relu(bias2 + sum_t w2_t . relu(w1 . x_t + bias1_t)) over the T+1 frames of `next`, bf16 MFMA.
extern "C" int fbbev_history_fused_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                                      const float* rt_flow, const float* w1, const float* bias1, const float* w2,
                                      const float* bias2, int B, int T, int C, int Cout, int Z, int Y, int X, float* out,
                                      void* workspace, size_t workspace_bytes, int elem_type, fbbev_stream_t stream_) {
    if (B < 0 || T <= 0 || C <= 0 || Cout <= 0 || Z <= 0 || Y <= 0 || X <= 0) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!history || !next || !rt_flow || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (C != 80 || Cout != 80 || (elem_type != 1 && elem_type != 2) || Z < 2 || Y < 2 || X < 2) return FBBEV_E_UNSUPPORTED;
    const long long N = (long long)Z * Y * X, frame = N * C;
    const int T1 = T + 1;
    if (history_stride_b == 0) history_stride_b = (long long)T * frame;
    if (next_stride_b == 0) next_stride_b = (long long)T1 * frame;
    if (history_stride_b < (long long)T * frame || next_stride_b < (long long)T1 * frame) return FBBEV_E_BADARG;
    if (history_stride_b % 8 != 0 || next_stride_b % 8 != 0 || !aligned16(history) || !aligned16(next) || !aligned16(bias1))
        return FBBEV_E_UNSUPPORTED;
    if (frame * 2 >= (1ll << 32) || N >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;              // 32-bit byte offsets inside a frame
    constexpr int MT = 5, KS = 3;
    const size_t need = ((size_t)MT * KS + (size_t)T1 * MT * KS) * 64 * 8 * sizeof(unsigned short);
    if (!workspace || !aligned16(workspace) || workspace_bytes < need) return FBBEV_E_WORKSPACE;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    unsigned short* w1f = static_cast<unsigned short*>(workspace);
    unsigned short* w2f = w1f + (size_t)MT * KS * 64 * 8;
    const int nfrag = (MT * KS + T1 * MT * KS) * 64;
    FBBEV_LAUNCH(k_history_weight_fragments_bf16, (nfrag + 255) / 256, 256, 0, stream, w1, w2, MT, MT, C, T1, w1f);
    FBBEV_CHECK_LAUNCH();
    const int n_xc = (X + 63) / 64;                               // 64-voxel tiles per grid row
    int YB = 128 / Z;                                             // (z, y) slabs per x chunk, as the warp kernel orders its workgroups
    if (YB < 1) YB = 1;
    if (YB > Y) YB = Y;
    const int nyb = (Y + YB - 1) / YB;
    const long long blocks = (long long)B * nyb * n_xc * YB * Z;
    if (blocks >= (1ll << 31) - 8) return FBBEV_E_UNSUPPORTED;
    const int per_xcd = (int)((blocks + 7) / 8);
    constexpr int A2S = ((MT * KS * 64 + 255) / 256) * 256 * 8;
    // 4 W2 buffers + the W1 fragments + the Y rows + 4 operand tiles: 137 KB (one 1024-thread workgroup per CU)
    const size_t lds = ((size_t)4 * A2S + (size_t)MT * KS * 64 * 8 + (size_t)4 * 16 * (KS * 32 + 8) + (size_t)4 * 64 * FBBEV_HF_XP) *
                       sizeof(unsigned short) + (size_t)4 * 80 * sizeof(float);
#define FBBEV_HF(ET_)                                                                                                  \
    do {                                                                                                              \
        int e_ = fbbev_rt_allow_dyn_lds((const void*)k_history_fused_bf16<ET_>, lds);                                  \
        if (e_) return e_;                                                                                            \
        FBBEV_LAUNCH((k_history_fused_bf16<ET_>), (long long)per_xcd * 8, 1024, lds, stream, history, history_stride_b, \
                     next, next_stride_b, rt_flow, (const unsigned short*)w1f, bias1, (const unsigned short*)w2f, bias2, \
                     T1, Z, Y, X, n_xc, YB, nyb, per_xcd, (int)blocks, out);                                           \
    } while (0)
    if (elem_type == 1) FBBEV_HF(1); else FBBEV_HF(2);
#undef FBBEV_HF
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// the fp32-MFMA convolutions on a voxel-major ring (feats (B, T1, N, C)); C = Cout in {16, 80}, workspace required
extern "C" int fbbev_history_conv_vm(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                     const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                     float* out, void* workspace, size_t workspace_bytes, int elem_type,
                                     fbbev_stream_t stream_) {
    if (B < 0 || T1 <= 0 || C <= 0 || Cout <= 0 || N < 0 || elem_type < 0 || elem_type > 2) return FBBEV_E_BADARG;
    if (B == 0 || N == 0) return 0;
    if (!feats || !w1 || !bias1 || !w2 || !bias2 || !out) return FBBEV_E_BADARG;
    if (!((C == 80 && Cout == 80) || (C == 16 && Cout == 16))) return FBBEV_E_UNSUPPORTED;
    if (!workspace) return FBBEV_E_WORKSPACE;
    if (feats_stride_b == 0) feats_stride_b = (long long)T1 * C * N;
    if (feats_stride_b < (long long)T1 * C * N) return FBBEV_E_BADARG;
    if (!aligned16(feats) || feats_stride_b % 8 != 0) return FBBEV_E_UNSUPPORTED;                      // 8- / 16-byte row pieces
    if ((long long)N * C * (elem_type == 0 ? 4 : 2) >= (1ll << 32)) return FBBEV_E_UNSUPPORTED;        // 32-bit byte offsets in a frame
    if (!aligned16(bias1)) return FBBEV_E_UNSUPPORTED;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    if (elem_type == 0) return history_conv_launch<0, true>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
    if (elem_type == 1) return history_conv_launch<1, true>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
    return history_conv_launch<2, true>(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace, workspace_bytes, stream);
}

extern "C" int fbbev_history_conv(const float* feats, long long feats_stride_b, const float* w1, const float* bias1,
                                  const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N,
                                  float* out, void* workspace, size_t workspace_bytes, fbbev_stream_t stream_) {
    return fbbev_history_conv_e(feats, feats_stride_b, w1, bias1, w2, bias2, B, T1, C, Cout, N, out, workspace,
                                workspace_bytes, 0, stream_);
}

static int conv3d_launch(const float* x, const float* weight_fragments, const float* bias, const float* residual, int B,
                         int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksize, int stride, int pad,
                         int relu, int mode, float* out, fbbev_stream_t stream_, bool planar = false) {
    if (B == 0) return 0;
    if (!x || !weight_fragments || !bias || !out) return FBBEV_E_BADARG;
    if (Cin % 16 != 0 || !aligned16(x) || !aligned16(weight_fragments) || !aligned16(bias) || !aligned16(out) ||
        (residual && !aligned16(residual))) return FBBEV_E_UNSUPPORTED;
    const long long nvox = (long long)B * Do * Ho * Wo;
    const long long gx = (nvox + 255) / 256;
    const int mt_total = (Cout + 15) / 16;
    const int MT = mt_total % 4 == 0 ? 4 : (mt_total % 2 == 0 ? 2 : 1);
    const long long pstride = (long long)(Cin / 16) * mt_total * 256;
    const int gy = mt_total / MT;
    const long long grid = gx * gy * (mode == 1 ? 8 : 1);
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
#define FBBEV_CONV3D(KD_, KS_, MT_)                                                                                      \
    FBBEV_LAUNCH((k_conv3d_ndhwc<KD_, KS_, MT_>), grid, 256, 0, (fbbev_rt_stream)stream_, x, weight_fragments, bias,  \
                 residual, out, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, mt_total, stride, pad, relu ? 1 : 0, mode, pstride, \
                 (int)gx, gy)
#define FBBEV_CONV3D_K(KD_, KS_) do { if (MT == 4) FBBEV_CONV3D(KD_, KS_, 4); else if (MT == 2) FBBEV_CONV3D(KD_, KS_, 2); else FBBEV_CONV3D(KD_, KS_, 1); } while (0)
    if (ksize == 3 && planar) FBBEV_CONV3D_K(1, 3); else if (ksize == 3) FBBEV_CONV3D_K(3, 3); else if (ksize == 2) FBBEV_CONV3D_K(2, 2);
    else FBBEV_CONV3D_K(1, 1);
#undef FBBEV_CONV3D_K
#undef FBBEV_CONV3D
    FBBEV_CHECK_LAUNCH();
    return 0;
}

static bool conv3d_geometry_ok(int in, int out, int ksize, int stride, int pad) { return out == (in + 2 * pad - ksize) / stride + 1; }

extern "C" int fbbev_conv3d_ndhwc(const float* x, const float* weight_fragments, const float* bias, const float* residual,
                                  int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksize,
                                  int stride, int pad, int relu, int transposed, float* out, fbbev_stream_t stream_) {
    if (B < 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if (transposed) {
        if (Do != Di || Ho != Hi || Wo != Wi) return FBBEV_E_BADARG;
        ksize = 1; stride = 1; pad = 0;
    } else {
        if (ksize < 1 || ksize > 3 || (stride != 1 && stride != 2) || pad < 0 || pad > 1) return FBBEV_E_UNSUPPORTED;
        if (!conv3d_geometry_ok(Di, Do, ksize, stride, pad) || !conv3d_geometry_ok(Hi, Ho, ksize, stride, pad) ||
            !conv3d_geometry_ok(Wi, Wo, ksize, stride, pad)) return FBBEV_E_BADARG;
    }
    return conv3d_launch(x, weight_fragments, bias, residual, B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, ksize, stride, pad, relu,
                         transposed ? 1 : 0, out, stream_);
}

extern "C" int fbbev_conv2d_nhwc(const float* x, const float* weight_fragments, const float* bias, const float* residual,
                                 int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad,
                                 int relu, float* out, fbbev_stream_t stream_) {
    if (B < 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || pad < 0 || pad > 1) return FBBEV_E_UNSUPPORTED;
    if (!conv3d_geometry_ok(Hi, Ho, ksize, stride, pad) || !conv3d_geometry_ok(Wi, Wo, ksize, stride, pad)) return FBBEV_E_BADARG;
    // an NHWC image is an NDHWC volume with one plane; the planar instantiation has a single tap along that axis
    return conv3d_launch(x, weight_fragments, bias, residual, B, 1, Hi, Wi, Cin, 1, Ho, Wo, Cout, ksize, stride, pad, relu, 0, out,
                         stream_, true);
}

extern "C" int fbbev_conv3d_ndhwc_bf16(const float* x, const void* weight_fragments_bf16, const float* bias,
                                       const float* residual, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo,
                                       int Cout, int ksize, int stride, int pad, int relu, int transposed, int planar,
                                       float* out, fbbev_stream_t stream_) {
    if (B < 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if (transposed) {
        if (Do != Di || Ho != Hi || Wo != Wi || planar) return FBBEV_E_BADARG;
        ksize = 1; stride = 1; pad = 0;
    } else {
        if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || pad < 0 || pad > 1) return FBBEV_E_UNSUPPORTED;
        if (planar ? (Di != 1 || Do != 1) : !conv3d_geometry_ok(Di, Do, ksize, stride, pad)) return FBBEV_E_BADARG;
        if (!conv3d_geometry_ok(Hi, Ho, ksize, stride, pad) || !conv3d_geometry_ok(Wi, Wo, ksize, stride, pad)) return FBBEV_E_BADARG;
    }
    if (B == 0) return 0;
    if (!x || !weight_fragments_bf16 || !bias || !out) return FBBEV_E_BADARG;
    if (Cin % 32 != 0 || !aligned16(x) || !aligned16(weight_fragments_bf16) || !aligned16(bias) || !aligned16(out) ||
        (residual && !aligned16(residual))) return FBBEV_E_UNSUPPORTED;
    const long long nvox = (long long)B * Do * Ho * Wo;
    const long long gx = (nvox + 255) / 256;
    const int mt_total = (Cout + 15) / 16;
    const int MT = mt_total % 4 == 0 ? 4 : (mt_total % 2 == 0 ? 2 : 1);
    const long long pstride = (long long)(Cin / 32) * mt_total * 512;
    const int gy = mt_total / MT;
    const long long grid = gx * gy * (transposed ? 8 : 1);
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const unsigned short* wfb = static_cast<const unsigned short*>(weight_fragments_bf16);
#define FBBEV_CONV3DB(KD_, KS_, MT_)                                                                                 \
    FBBEV_LAUNCH((k_conv3d_ndhwc_bf16<KD_, KS_, MT_>), grid, 256, 0, (fbbev_rt_stream)stream_, x, wfb, bias, residual, out, \
                 B, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, mt_total, stride, pad, relu ? 1 : 0, transposed ? 1 : 0, pstride,  \
                 (int)gx, gy)
#define FBBEV_CONV3DB_K(KD_, KS_) do { if (MT == 4) FBBEV_CONV3DB(KD_, KS_, 4); else if (MT == 2) FBBEV_CONV3DB(KD_, KS_, 2); else FBBEV_CONV3DB(KD_, KS_, 1); } while (0)
    if (ksize == 3 && planar) FBBEV_CONV3DB_K(1, 3); else if (ksize == 3) FBBEV_CONV3DB_K(3, 3); else FBBEV_CONV3DB_K(1, 1);
#undef FBBEV_CONV3DB_K
#undef FBBEV_CONV3DB
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_conv3d_k3s1_tiled_bf16(const float* x, const void* weight_fragments_bf16, const float* bias,
                                            const float* residual, int B, int D, int H, int W, int Cin, int Cout, int relu,
                                            float* out, fbbev_stream_t stream_) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!x || !weight_fragments_bf16 || !bias || !out) return FBBEV_E_BADARG;
    if (Cin % 32 != 0 || !aligned16(x) || !aligned16(weight_fragments_bf16) || !aligned16(bias) || !aligned16(out) ||
        (residual && !aligned16(residual))) return FBBEV_E_UNSUPPORTED;
    const int tiles_d = (D + 3) / 4, tiles_h = (H + 7) / 8, tiles_w = (W + 7) / 8;
    const int mt_total = (Cout + 15) / 16;
    const int MT = mt_total % 4 == 0 ? 4 : (mt_total % 2 == 0 ? 2 : 1);
    const int gy = mt_total / MT;
    const long long n_tiles = (long long)B * tiles_d * tiles_h * tiles_w;
    const long long grid = (n_tiles + 7) / 8 * 8 * gy;          // whole groups of 8 tiles (one per XCD), see the kernel
    if (grid >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * 600 * 32 * sizeof(unsigned short);
    const unsigned short* wfb = static_cast<const unsigned short*>(weight_fragments_bf16);
#define FBBEV_CONV3DT(MT_)                                                                                           \
    do {                                                                                                             \
        int e = fbbev_rt_allow_dyn_lds((const void*)k_conv3d_k3_tile_bf16<MT_>, lds);                                \
        if (e) return e;                                                                                             \
        FBBEV_LAUNCH((k_conv3d_k3_tile_bf16<MT_>), grid, 512, lds, (fbbev_rt_stream)stream_, x, wfb, bias, residual, out, B, \
                     D, H, W, Cin, Cout, mt_total, relu ? 1 : 0, tiles_d, tiles_h, tiles_w, gy);                      \
    } while (0)
    if (MT == 4) FBBEV_CONV3DT(4); else if (MT == 2) FBBEV_CONV3DT(2); else FBBEV_CONV3DT(1);
#undef FBBEV_CONV3DT
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_conv3d_dgrad_ndhwc(const float* dy, const float* weight_fragments_t, const float* zero_bias, int B,
                                        int Do, int Ho, int Wo, int Cout, int Di, int Hi, int Wi, int Cin, int ksize,
                                        int stride, int pad, float* dx, fbbev_stream_t stream_) {
    if (B < 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if (ksize < 1 || ksize > 3 || (stride != 1 && stride != 2) || pad < 0 || pad > 1) return FBBEV_E_UNSUPPORTED;
    if (!conv3d_geometry_ok(Di, Do, ksize, stride, pad) || !conv3d_geometry_ok(Hi, Ho, ksize, stride, pad) ||
        !conv3d_geometry_ok(Wi, Wo, ksize, stride, pad)) return FBBEV_E_BADARG;
    // gather form: the kernel's "input" is dy (Cout channels), its "output" voxels are those of dx (Cin channels)
    return conv3d_launch(dy, weight_fragments_t, zero_bias, nullptr, B, Do, Ho, Wo, Cout, Di, Hi, Wi, Cin, ksize, stride, pad, 0,
                         2, dx, stream_);
}

extern "C" int fbbev_conv3d_wgrad_ndhwc(const float* x, const float* dy, int B, int Di, int Hi, int Wi, int Cin, int Do,
                                        int Ho, int Wo, int Cout, int ksize, int stride, int pad, float* dw,
                                        fbbev_stream_t stream_) {
    if (B < 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return FBBEV_E_BADARG;
    if (ksize < 1 || ksize > 3 || (stride != 1 && stride != 2) || pad < 0 || pad > 1) return FBBEV_E_UNSUPPORTED;
    if (!conv3d_geometry_ok(Di, Do, ksize, stride, pad) || !conv3d_geometry_ok(Hi, Ho, ksize, stride, pad) ||
        !conv3d_geometry_ok(Wi, Wo, ksize, stride, pad)) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!x || !dy || !dw) return FBBEV_E_BADARG;
    if (Cin % 4 != 0 || Cout % 4 != 0 || !aligned16(x) || !aligned16(dy)) return FBBEV_E_UNSUPPORTED;
    const long long nvox = (long long)B * Do * Ho * Wo;
    // voxel chunk per wave: enough chunks to fill the chip a few times over, at least 256 voxels (16 loop iterations)
    const int cout_blocks = (Cout + 63) / 64, cin_blocks = (Cin + 63) / 64, T = ksize * ksize * ksize;
    long long chunk = nvox / 64;
    if (chunk < 256) chunk = 256;
    if (chunk > 4096) chunk = 4096;
    chunk = (chunk + 15) / 16 * 16;
    const long long n_chunks = (nvox + chunk - 1) / chunk;
    const long long tasks = n_chunks * cout_blocks * cin_blocks * T;
    const long long blocks = (tasks + 3) / 4;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
#define FBBEV_WGRAD(KS_)                                                                                             \
    FBBEV_LAUNCH((k_conv3d_wgrad_ndhwc<KS_>), blocks, 256, 0, (fbbev_rt_stream)stream_, x, dy, dw, B, Di, Hi, Wi, Cin, Do, \
                 Ho, Wo, Cout, stride, pad, (int)chunk, (int)n_chunks, cout_blocks, cin_blocks)
    if (ksize == 3) FBBEV_WGRAD(3); else if (ksize == 2) FBBEV_WGRAD(2); else FBBEV_WGRAD(1);
#undef FBBEV_WGRAD
    FBBEV_CHECK_LAUNCH();
    return 0;
}

extern "C" int fbbev_blend_levels_ndhwc(const float* level0, const float* const* coarse, const int* coarse_dims,
                                        int n_coarse, const float* wsoft, int K, int B, int D, int H, int W, int C,
                                        float* out, fbbev_stream_t stream_) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || n_coarse < 0 || n_coarse > 3 || K < n_coarse + 1) return FBBEV_E_BADARG;
    if (B == 0) return 0;
    if (!level0 || !wsoft || !out || (n_coarse > 0 && (!coarse || !coarse_dims))) return FBBEV_E_BADARG;
    if (C % 4 != 0 || !aligned16(level0) || !aligned16(out)) return FBBEV_E_UNSUPPORTED;
    fbbev_blend_level lv[3] = {{nullptr, 1, 1, 1}, {nullptr, 1, 1, 1}, {nullptr, 1, 1, 1}};
    for (int k = 0; k < n_coarse; ++k) {
        if (!coarse[k] || coarse_dims[3 * k] <= 0 || coarse_dims[3 * k + 1] <= 0 || coarse_dims[3 * k + 2] <= 0) return FBBEV_E_BADARG;
        if (!aligned16(coarse[k])) return FBBEV_E_UNSUPPORTED;
        lv[k] = fbbev_blend_level{coarse[k], coarse_dims[3 * k], coarse_dims[3 * k + 1], coarse_dims[3 * k + 2]};
    }
    const long long total = (long long)B * D * H * W * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    FBBEV_LAUNCH(k_blend_levels_ndhwc, blocks, 256, 0, (fbbev_rt_stream)stream_, level0, lv[0], lv[1], lv[2], n_coarse, wsoft, K,
                 B, D, H, W, C, out);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// the training-path entries live in their own translation unit (capi_train.hip: a header-only edit there recompiles in seconds); the
// CPU emulator build (tests/emu) is ONE translation unit and takes them from here
#ifdef FBBEV_TEST_OVERRIDES
#include "capi_train.hip"
#endif
