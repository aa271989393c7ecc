/*
 * fbbev.h -- C ABI of libfbbev_hip.so: the MI355X (gfx950) native view-transformation hot path
 * of FB-OCC (forward lift-splat + backward-projection sampling).
 *
 * Drop-in boundary.  Every entry point takes plain device pointers, sizes and a HIP stream;
 * nothing here knows about torch.  Each function names the reference interface it replaces
 * (paths relative to the NVlabs/FB-BEV tree).  All functions are re-entrant, hold no global
 * mutable state, never synchronise the host and enqueue on the caller's stream (the reference
 * launches on the legacy default stream, bev_pool_cuda.cu:124,133).
 *
 * Return value: 0 = ok, <0 = invalid argument (FBBEV_E_*), >0 = hipError_t of the failed launch.
 * Nothing throws across this ABI.
 */
#ifndef FBBEV_H_
#define FBBEV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fbbev_stream_t; /* hipStream_t */

#define FBBEV_E_BADARG (-1)
#define FBBEV_E_UNSUPPORTED (-2)
#define FBBEV_E_WORKSPACE (-3)

/* flags of fbbev_bev_pool_v2_dense_fwd (tuning knobs: none changes the result bits; OUT_BF16 / OUT_F16 select the
 * storage type of `out` -- the sums stay fp32 in-order fmaf chains and are rounded once at the store) */
#define FBBEV_POOL_STORE_MASK 0x3   /* output store cache policy: 0 plain, 1 nontemporal; with bit 17 set: sc1 nt */
#define FBBEV_POOL_CPL8 0x4         /* 8 channels per lane instead of 4 */
#define FBBEV_POOL_CSPLIT_SHIFT 4   /* bits 4-7: split the channel range over this many workgroups */
#define FBBEV_POOL_WG_SHIFT 8       /* bits 8-9: workgroup size 0 -> 256, 1 -> 128 threads */
#define FBBEV_POOL_XCD_SWIZZLE 0x400 /* deal chunks of consecutive tiles round-robin to the 8 XCDs */
#define FBBEV_POOL_STORE_HI_SHIFT 17 /* bit 17: `sc1 nt` stores (the default; do not evict the gathered inputs from L2) */
#define FBBEV_POOL_CHANNELS_LAST 0x100000 /* out is (B,Z,Y,X,C) -- the reference op's own layout -- written densely */
#define FBBEV_POOL_OUT_BF16 0x800000  /* `out` holds bfloat16: fp32 sums rounded once (nearest-even) at the store */
#define FBBEV_POOL_OUT_F16 0x1000000  /* `out` holds IEEE half; both: (B,C,Z,Y,X) layout only, (Y*X) % 8 == 0 */
#define FBBEV_POOL_SWZ_CHUNK_SHIFT 12 /* bits 12-16: log2(tiles per chunk) for the swizzle, 0 = default */
#define FBBEV_POOL_PIPE 0x4000000 /* fbbev_bev_pool_v2_dense_fwd[_add], fp32 volume, 64- / 128- / 256-voxel tiles: a workgroup walks a run
                                   * of consecutive tiles and keeps the NEXT tile's interval metadata / point indices in flight under the
                                   * current tile's gathers (k_pool_fwd_dense_pipe) -- the same bits; pays on dense grids (shipped config) */
#define FBBEV_POOL_GATHER8 0x8000000 /* experiment knob of fbbev_bev_pool_v2_dense_fwd[_add] (round 5), fp32 volume, 64- / 128-voxel tiles, 256 threads:
                                        eight points per gather batch instead of four (long intervals of dense grids); same bits */
#define FBBEV_POOL_SPLIT_LONG 0x2000000 /* opt-in TOLERANCE mode of fbbev_bev_pool_v2_dense_fwd[_add]: an interval of more than 32
                                         * points is summed by up to 32 lane groups of its workgroup (contiguous chunks in order,
                                         * partial sums added in group order: deterministic) -- equal to the reference's serial
                                         * chain (bev_pool_cuda.cu:33-38) up to fp32 reassociation (<= 1e-4, north_star's bar),
                                         * not bit for bit.  fp32 volume, sc1-nt stores, 256 threads, tile_voxels 64 / 128; the
                                         * default (flag clear) stays bit-exact */

int fbbev_version(void);

/* ----------------------------------------------------------------------------------------------
 * Boundary 1: mmdet3d.ops.bev_pool_v2.bev_pool_v2_ext
 * -------------------------------------------------------------------------------------------- */

/* Replaces  void bev_pool_v2(int c, int n_intervals, const float* depth, const float* feat,
 *   const int* ranks_depth, const int* ranks_feat, const int* ranks_bev,
 *   const int* interval_starts, const int* interval_lengths, float* out)
 *   -- mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:122-128 (kernel :18-45), reached from
 *   bev_pool_v2_forward, src/bev_pool.cpp:28-55.
 * depth (B,N,D,H,W) f32, feat (B,N,H,W,C) f32, out (B,Z,Y,X,C) f32 PRE-ZEROED by the caller
 * (bev_pool.py:24); only rows ranks_bev[interval_starts[i]] are written.  Each output element is
 * an fmaf chain over the interval in the given order == the reference kernel's arithmetic. */
int fbbev_bev_pool_v2_fwd(int c, int n_intervals, const float* depth, const float* feat,
                          const int32_t* ranks_depth, const int32_t* ranks_feat,
                          const int32_t* ranks_bev, const int32_t* interval_starts,
                          const int32_t* interval_lengths, float* out, fbbev_stream_t stream);

/* Replaces  void bev_pool_v2_grad(int c, int n_intervals, const float* out_grad, ...)
 *   -- src/bev_pool_cuda.cu:130-137 (kernel :64-118), reached from bev_pool_v2_backward,
 *   src/bev_pool.cpp:72-102.  Rank arrays are sorted by ranks_feat and the intervals are over
 *   ranks_feat (bev_pool.py:44-54).  depth_grad / feat_grad are pre-zeroed by the caller. */
int fbbev_bev_pool_v2_bwd(int c, int n_intervals, const float* out_grad, const float* depth,
                          const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                          const int32_t* ranks_bev, const int32_t* interval_starts,
                          const int32_t* interval_lengths, float* depth_grad, float* feat_grad,
                          fbbev_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Fused forward-projection entry points (additive; same arithmetic, no host sync)
 * -------------------------------------------------------------------------------------------- */

/* Replaces LSSViewTransformerFunction3D.get_lidar_coor
 *   -- fbbev/view_transformation/forward_projection/view_transformer.py:458-498.
 * xs (W), ys (H), ds (D): the three axes of the frustum template (create_frustum, :389-411);
 * rots/intrins/post_rots (B,N,3,3), trans/post_trans (B,N,3), bda (B,3,3); coor (B,N,D,H,W,3).
 * Closed-form 3x3 inverses: coor may differ from torch.inverse-based results in the last ulps
 * (the bit-exact contract starts at fbbev_rank_build's input). */
int fbbev_lidar_coor(const float* xs, const float* ys, const float* ds, const float* rots,
                     const float* trans, const float* intrins, const float* post_rots,
                     const float* post_trans, const float* bda, int B, int N, int D, int H, int W,
                     float* coor, fbbev_stream_t stream);

/* Replaces feat.permute(0,1,3,4,2) + .contiguous() (view_transformer.py:536, bev_pool.py:18):
 * in (n_images, C, HW) -> out (n_images, HW, C), f32, LDS-tiled transpose. */
int fbbev_nchw_to_nhwc(const float* in, float* out, int n_images, int C, int HW, fbbev_stream_t stream);

/* Camera tokens of the backward projection in one pass per feature level.  Replaces, in
 *   fbbev/view_transformation/backward_projection/bevformer_utils/bevformer.py:95-117
 * feat.flatten(3).permute(1,0,3,2) + cams_embeds[:,None,None,:], torch.cat over the levels and the
 * (num_cam, sum HW, bs, C) permute, and in spatial_cross_attention_depth.py:151,188-191 the rebatch
 * value.permute(2,0,1,3).reshape(bs*num_cams, sum HW, C):
 *   out[img*out_image_stride + out_offset + p*C + c] = in[img, c, p] (+ bias[(img % bias_rows)*C + c])
 * in (n_images, C, HW) f32; bias (bias_rows, C) or NULL (then exactly fbbev_nchw_to_nhwc with strides);
 * strides / offset in floats, out_image_stride >= C*HW.  The same call with the roles of C and HW
 * swapped is the inverse transposition (tokens -> channel planes). */
int fbbev_tokens_from_nchw(const float* in, float* out, int n_images, int C, int HW,
                           long long out_image_stride, long long out_offset, const float* bias,
                           int bias_rows, fbbev_stream_t stream);
/* The same transposition with a per-POSITION row added:  out[img, p, c] = in[img, c, p] + pos_bias[p, c]  (pos_bias (HW, C), the
 * same for every image): the BEV queries of the backward projection -- `lss_bev.flatten(2).permute(2, 0, 1)` + the learned
 * `bev_embedding` (backward_projection.py:96-99) -- in one pass. */
int fbbev_tokens_from_nchw_pos(const float* in, float* out, int n_images, int C, int HW, long long out_image_stride,
                               long long out_offset, const float* pos_bias, fbbev_stream_t stream);
/* fbbev_tokens_from_nchw for EVERY level of the camera-token pyramid in one launch (bevformer.py:95-117: per-level flatten / permute /
 * + cams_embeds, then torch.cat): in[l] is level l as (n_images, C, hw[l]) (HOST array of device pointers, n_levels <= 8); out is
 * (n_images, sum hw, C); bias as above. */
int fbbev_tokens_from_nchw_levels(const float* const* in, const int32_t* hw, int n_levels, float* out, int n_images, int C,
                                  const float* bias, int bias_rows, fbbev_stream_t stream);

/* Replaces LSSViewTransformerFunction3D.voxel_pooling_prepare_v2
 *   -- fbbev/view_transformation/forward_projection/view_transformer.py:547-605
 *   (~17 torch launches, an argsort and >=4 host syncs in the reference).
 * coor (B,N,D,H,W,3) f32 ego-frame frustum points.  lower/interval/grid_size are the three fp32
 * values of grid_lower_bound / grid_interval / grid_size (view_transformer.py:384-387).
 * Outputs are sized to the upper bound n = B*N*D*H*W; the valid prefix lengths are written to
 * counts[0] = P (points kept) and counts[1] = I (non-empty voxels) ON THE DEVICE.
 * Bit-exact with the reference for ranks_bev / interval_starts / interval_lengths, including the
 * fp32 rank evaluation and the truncation toward zero; (ranks_depth, ranks_feat) come out in the
 * canonical STABLE order (ascending point id inside a voxel).
 * interval_rank (optional, may be NULL) receives ranks_bev[interval_starts[i]]. */
size_t fbbev_rank_workspace_bytes(int64_t n_points);
int fbbev_rank_build(const float* coor, int B, int N, int D, int H, int W, const float* lower3,
                     const float* interval3, const float* grid_size3, int32_t* ranks_bev,
                     int32_t* ranks_depth, int32_t* ranks_feat, int32_t* interval_starts,
                     int32_t* interval_lengths, int32_t* interval_rank, int32_t* counts,
                     void* workspace, size_t workspace_bytes, fbbev_stream_t stream);

/* fbbev_rank_build with the BEVDet-era point filter  kept &= depth.view(-1) > depth_threshold
 *   -- mmdet3d/models/necks/view_transformer.py:552-557 (0.01 there): `depth` is the (B,N,D,H,W) depth distribution the
 * pooling will read; points whose own depth probability does not exceed the threshold are dropped before ranking, so
 * P becomes data dependent -- the device-side counts absorb that without a host sync. */
int fbbev_rank_build_depth(const float* coor, const float* depth, float depth_threshold, int B, int N, int D, int H,
                           int W, const float* lower3, const float* interval3, const float* grid_size3,
                           int32_t* ranks_bev, int32_t* ranks_depth, int32_t* ranks_feat, int32_t* interval_starts,
                           int32_t* interval_lengths, int32_t* interval_rank, int32_t* counts, void* workspace,
                           size_t workspace_bytes, fbbev_stream_t stream);

/* get_lidar_coor + voxel_pooling_prepare_v2 in one call (view_transformer.py:458-498 + :547-605): the
 * keys are evaluated from the camera parameters inside the sort's first pass -- the same per-point
 * arithmetic as fbbev_lidar_coor followed by fbbev_rank_build, hence the same index tensors -- and
 * `coor` is never materialised.  Arguments as in those two functions; `frustum` is the optional
 * (D,H,W,3) template of create_frustum (view_transformer.py:389-411) -- when given, points are looked up
 * in it instead of being decomposed with integer divisions (same values, it is stacked from xs/ys/ds). */
int fbbev_lift_rank_build(const float* frustum, const float* xs, const float* ys, const float* ds,
                          const float* rots,
                          const float* trans, const float* intrins, const float* post_rots,
                          const float* post_trans, const float* bda, int B, int N, int D, int H, int W,
                          const float* lower3, const float* interval3, const float* grid_size3,
                          int32_t* ranks_bev, int32_t* ranks_depth, int32_t* ranks_feat,
                          int32_t* interval_starts, int32_t* interval_lengths, int32_t* interval_rank,
                          int32_t* counts, void* workspace, size_t workspace_bytes, fbbev_stream_t stream);

/* Camera-parameter-keyed index cache (the reference's `pre_compute` / `init_acceleration_v2`,
 * view_transformer.py:500-519,607-611, which upstream disables with `assert False` at :628 because nothing
 * invalidates it): the index tensors depend only on the six camera tensors.  fbbev_lift_rank_build_cached first
 * compares their bits ON THE DEVICE with `cam_key` (fbbev_cam_key_words(B,N) uint32, caller-owned, initialise to
 * 0xFFFFFFFF): equal -> cache_state[0] = 1 and every kernel of the build returns at once, the index tensors / counts of
 * the previous call stay valid (the caller passes the SAME buffers every time); different -> the key is refreshed,
 * cache_state[0] = 0, cache_state[1] += 1 (number of builds) and the build runs.  No host sync, graph-capturable.
 * fbbev_pool_tile_index_cached is the tile index with the same early-out -- per TABLE: `table_gate` (2 x int32, caller-owned,
 * one pair per tile table, initialise to -1) remembers the build number (cache_state[1]) the table was built for; the
 * table is kept only when the index set is unchanged AND this very table was built for it, otherwise it is rebuilt
 * (a table of another tile size, or one allocated after the last build, is never trusted).  One extra 1-thread launch. */
size_t fbbev_cam_key_words(int B, int N);
int fbbev_lift_rank_build_cached(const float* frustum, const float* xs, const float* ys, const float* ds,
                                 const float* rots, const float* trans, const float* intrins,
                                 const float* post_rots, const float* post_trans, const float* bda, int B, int N,
                                 int D, int H, int W, const float* lower3, const float* interval3,
                                 const float* grid_size3, int32_t* ranks_bev, int32_t* ranks_depth,
                                 int32_t* ranks_feat, int32_t* interval_starts, int32_t* interval_lengths,
                                 int32_t* interval_rank, int32_t* counts, void* workspace, size_t workspace_bytes,
                                 uint32_t* cam_key, int32_t* cache_state, fbbev_stream_t stream);
int fbbev_pool_tile_index_cached(const int32_t* interval_rank, const int32_t* interval_starts,
                                 const int32_t* counts, int n_intervals_max, int B, int Z, int Y, int X,
                                 int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                                 const int32_t* cache_state, int32_t* table_gate, fbbev_stream_t stream);

/* Fused replacement of  feat.new_zeros + bev_pool_v2_forward + permute(0,4,1,2,3).contiguous()
 *   -- bev_pool.py:24-35,88.  Two launches:
 * fbbev_pool_tile_index: for every tile of `tile_voxels` (64..1024) consecutive voxels of a (b,z)
 *   plane, the first interval whose rank falls in it (parallel lower bound over interval_rank) and
 *   that interval's first point.  counts = device-side [P, I] of fbbev_rank_build: no host sync.
 * fbbev_bev_pool_v2_dense_fwd: writes EVERY element of out (B,C,Z,Y,X) exactly once (zeros for empty
 *   voxels) in the final layout, so `out` need not be pre-zeroed.  Same in-order fmaf chains as
 *   fbbev_bev_pool_v2_fwd => identical bits.  interval_rank[i] = ranks_bev[interval_starts[i]].
 *   Requires C % 4 == 0, (Y*X) % 4 == 0, 16-byte aligned feat/out (else FBBEV_E_UNSUPPORTED; callers
 *   fall back to fbbev_bev_pool_v2_fwd).
 *   out_stride_b / out_stride_c: element strides of the batch and channel dimensions of `out`
 *   (0 = contiguous); the (Z,Y,X) block of one channel is always contiguous.
 *   With FBBEV_POOL_OUT_BF16 / FBBEV_POOL_OUT_F16 `out` points to 16-bit elements (strides in those elements,
 *   multiples of 8; (Y*X) % 8 == 0): the storage formats BASELINE configs[1] (bf16) and configs[4] (fp16) name.
 *   With FBBEV_POOL_CHANNELS_LAST (pass the same flag to fbbev_pool_tile_index) `out` is instead the
 *   reference op's own (B,Z,Y,X,C) layout (bev_pool.py:24), every row written once: the whole launch is
 *   one linear store stream; callers take out.permute(0,4,1,2,3) as a view instead of copying (:88).
 *   Channels-last tile_voxels of 8/16/32 select the "one 16-byte store per thread" kernel (a workgroup
 *   writes tile_voxels*C*4 contiguous bytes, e.g. 5 KiB at C=80) -- the shape that reaches the HBM write
 *   ceiling on MI355X.
 * tile_ws: fbbev_pool_dense_workspace_bytes(B,Z,Y,X) bytes, shared by the two calls. */
size_t fbbev_pool_dense_workspace_bytes(int B, int Z, int Y, int X);
int fbbev_pool_tile_index(const int32_t* interval_rank, const int32_t* interval_starts,
                          const int32_t* counts, int n_intervals_max, int B, int Z, int Y, int X,
                          int tile_voxels, int flags, void* tile_ws, size_t tile_ws_bytes,
                          fbbev_stream_t stream);
int fbbev_bev_pool_v2_dense_fwd(const float* depth, const float* feat, const int32_t* ranks_depth,
                                const int32_t* ranks_feat, const int32_t* interval_rank,
                                const int32_t* interval_starts, const int32_t* interval_lengths, int B,
                                int C, int Z, int Y, int X, float* out_bczyx, long long out_stride_b,
                                long long out_stride_c, const void* tile_ws, size_t tile_ws_bytes,
                                int tile_voxels, int flags, fbbev_stream_t stream);

/* The forward + backward projection of FB-OCC needs the pooled volume twice -- its Z-mean as the backward projection's
 * input (fbocc.py:359 `lss_bev=bev_feat.mean(-1)`) and, after the refinement, re-added to it (:365-366
 * `bev_feat_refined[..., None] + bev_feat`): the reference writes the volume, reads it for the mean, and reads +
 * re-writes it for the add.  With these two entry points the volume is written once and never re-read:
 * fbbev_pool_zmean: out_mean (B,C,Y,X) = mean over z of the pooled sums, straight from the index tensors (planes
 *   accumulated in ascending z, same in-order fmaf chains per voxel); same tile index / tile_voxels (64..256) / csplit
 *   and CPL8 flags as the dense kernel.
 * fbbev_bev_pool_v2_dense_fwd_add: fbbev_bev_pool_v2_dense_fwd with out[b,c,z,y,x] = pooled + addend[b,c,y,x]
 *   (addend (B,C,Y,X) f32 contiguous, 16-byte aligned; (B,C,Z,Y,X) layout only). */
int fbbev_pool_zmean(const float* depth, const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                     const int32_t* interval_rank, const int32_t* interval_starts, const int32_t* interval_lengths,
                     int B, int C, int Z, int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                     int tile_voxels, int flags, fbbev_stream_t stream);
/* fbbev_pool_zmean with the result written as the backward projection's QUERY ROWS: out_rows (B, Y*X, C) = mean over z + row_bias
 * (Y*X, C) (the module's bev_embedding; NULL: none) -- backward_projection.py:96-99's flatten(2).permute + `+ bev_embedding` done by
 * the Z-mean's store instead of a transposing pass over (B, C, Y, X); the same values (one fp32 add per element).  Single pass
 * only (the z_groups form keeps the planes layout). */
int fbbev_pool_zmean_rows(const float* depth, const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                          const int32_t* interval_rank, const int32_t* interval_starts, const int32_t* interval_lengths, int B,
                          int C, int Z, int Y, int X, const float* row_bias, float* out_rows, const void* tile_ws,
                          size_t tile_ws_bytes, int tile_voxels, int flags, fbbev_stream_t stream);
/* fbbev_pool_zmean with the Z planes of a tile dealt to z_groups workgroups (each walks ceil(Z / z_groups) planes) + a caller-owned
 * partial buffer of z_groups * B*C*Y*X floats; a second small kernel adds the groups in order and divides by Z.  For grids with
 * few tiles (the shipped 100x100x8 grid at small batch), where the single pass is one Z-plane latency chain per workgroup.
 * z_groups = Z (one plane per group) gives the single pass's bits (both add the per-plane sums of a pillar in z order); 1 < z_groups
 * < Z is another association of the z sum (fp32 rounding); z_groups = 1 is fbbev_pool_zmean. */
int fbbev_pool_zmean_split(const float* depth, const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                           const int32_t* interval_rank, const int32_t* interval_starts, const int32_t* interval_lengths,
                           int B, int C, int Z, int Y, int X, float* out_mean, const void* tile_ws, size_t tile_ws_bytes,
                           int tile_voxels, int flags, int z_groups, void* partial_ws, size_t partial_ws_bytes,
                           fbbev_stream_t stream);
int fbbev_bev_pool_v2_dense_fwd_add(const float* depth, const float* feat, const int32_t* ranks_depth,
                                    const int32_t* ranks_feat, const int32_t* interval_rank,
                                    const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C,
                                    int Z, int Y, int X, float* out, long long out_stride_b, long long out_stride_c,
                                    const void* tile_ws, size_t tile_ws_bytes, int tile_voxels, int flags,
                                    const float* addend, fbbev_stream_t stream);

/* The forward projection as ONE entry (SURVEY 8b `fbbev_lift_splat_fused`).  Replaces, in
 *   fbbev/view_transformation/forward_projection/view_transformer.py:521-545 (voxel_pooling_v2) + :613-635 (view_transform_core),
 * get_lidar_coor (:458-498) -> voxel_pooling_prepare_v2 (:547-605) -> feat.permute(0,1,3,4,2) (:536) -> bev_pool_v2 (bev_pool.py:83-89:
 * new_zeros + bev_pool_v2_forward + permute().contiguous()), i.e. exactly the sequence
 *   fbbev_lift_rank_build[_cached] -> fbbev_nchw_to_nhwc -> fbbev_pool_tile_index[_cached] -> fbbev_bev_pool_v2_dense_fwd
 * on the caller's stream, bit-identical to issuing the four calls (tests/test_gpu_parity.py), no host sync, graph-capturable.
 * depth (B,N,D,H,W) f32, context (B,N,C,H,W) f32 (the depth net's two outputs, NCHW as it leaves them); camera tensors, frustum
 * axes and grid triples as fbbev_lift_rank_build; out / strides / tile_voxels / flags as fbbev_bev_pool_v2_dense_fwd (the storage
 * flags select a 16-bit `out`).  workspace: fbbev_lift_splat_fused_ws_bytes(...) bytes, 16-byte aligned, caller-owned; it holds the
 * index tensors of the call, which a backward pass (fbbev_bev_pool_v2_dense_bwd) or a second consumer (fbbev_pool_zmean) reads at
 * the byte offsets fbbev_lift_splat_fused_ws_offsets reports: [ranks_bev, ranks_depth, ranks_feat, interval_starts,
 * interval_lengths, interval_rank, counts (int32 [P, I]), feat rows (B,N,H,W,C) f32].
 * cam_key / cache_state: both NULL = indices rebuilt every call (the reference's own setting, :628); both given = the camera-keyed
 * cache of fbbev_lift_rank_build_cached with the SAME workspace passed every call: cam_key fbbev_cam_key_words(B,N) uint32
 * initialised to 0xFFFFFFFF, cache_state FOUR int32 initialised to {0, 0, -1, -1} (hit flag, build count, tile-table gate pair). */
size_t fbbev_lift_splat_fused_ws_bytes(int B, int N, int D, int H, int W, int C, int Z, int Y, int X);
int fbbev_lift_splat_fused_ws_offsets(int B, int N, int D, int H, int W, int C, int Z, int Y, int X, size_t* offsets8);
int fbbev_lift_splat_fused(const float* frustum, const float* xs, const float* ys, const float* ds, const float* rots,
                           const float* trans, const float* intrins, const float* post_rots, const float* post_trans,
                           const float* bda, const float* depth, const float* context, int B, int N, int D, int H, int W, int C,
                           const float* lower3, const float* interval3, const float* grid_size3, int Z, int Y, int X, void* out,
                           long long out_stride_b, long long out_stride_c, int tile_voxels, int flags, void* workspace,
                           size_t workspace_bytes, uint32_t* cam_key, int32_t* cache_state, fbbev_stream_t stream);

/* Measurement aid, not part of the product path (bench.py `roofline.store_floor_ms` / `no_gather_ms`): the default fp32
 * instantiation of the dense kernel (tile_voxels 128, FBBEV_POOL_CPL8, 256 threads, `sc1 nt` stores; anything else ->
 * FBBEV_E_UNSUPPORTED) with its gathers compiled out, launched with the grid / tile walk / XCD order the product launch
 * takes for the same `flags`.  mode 1: the store pattern alone (every tile written as zeros, no metadata); mode 2:
 * everything except the depth / feature gathers and their fmaf chains; mode 3 (round 5): everything except the stores (what
 * the store stream has to hide; `out` untouched).  `out` (B,C,Z,Y,X) f32 contiguous receives zeros.  With FBBEV_POOL_OUT_BF16
 * in `flags` (round 5) the same three modes of the bf16-storage leg's instantiation (16-bit LDS tile, 128 voxels; `out` then
 * points to 16-bit elements). */
int fbbev_diag_pool_store_floor(const float* depth, const float* feat, const int32_t* ranks_depth,
                                const int32_t* ranks_feat, const int32_t* interval_rank,
                                const int32_t* interval_starts, const int32_t* interval_lengths, int B, int C, int Z,
                                int Y, int X, float* out, const void* tile_ws, size_t tile_ws_bytes, int tile_voxels,
                                int flags, int mode, fbbev_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Boundary 2: mmcv._ext.ms_deform_attn_{forward,backward} (mmcv-full 1.5.2, external to the tree)
 * -------------------------------------------------------------------------------------------- */

/* Replaces ext_module.ms_deform_attn_forward(value, spatial_shapes, level_start_index,
 *   sampling_locations, attention_weights, im2col_step=...)
 *   -- call sites bevformer_utils/multi_scale_deformable_attn_function.py:127-133,
 *   spatial_cross_attention_depth.py:586-588,593-595.
 * value (B,S,M,Dh) f32; spatial_shapes (L,2) int64 (h,w); level_start_index (L) int64;
 * sampling_loc (B,Q,M,L,P,2) f32 (x,y) normalised; attn_weight (B,Q,M,L,P) f32; out (B,Q,M*Dh).
 * im2col_step is accepted and ignored (any batch size works). */
int fbbev_msda_fwd(const float* value, const int64_t* spatial_shapes,
                   const int64_t* level_start_index, const float* sampling_loc,
                   const float* attn_weight, int batch, int spatial_size, int num_heads,
                   int channels, int num_levels, int num_query, int num_point, float* out,
                   fbbev_stream_t stream);

/* BEV self-attention sampling with the location arithmetic folded in (inference): replaces, inside mmcv's
 * MultiScaleDeformableAttention.forward, `reference_points[:, :, None, :, None, :] + sampling_offsets /
 * offset_normalizer` (two elementwise passes over (B,Q,M,L,P,2)) + ms_deform_attn_forward.  ref_points (B,Q,L,2);
 * offsets (B,Q,M,L,P,2) raw [offsets_head_minor: (B,Q,L,P,M,2)]; attn_weight (B,Q,M,L,P) softmaxed; value
 * (B,S,M,head_stride) with `channels` used floats per head (0 = dense).  channels in {4,8,10,16,32}. */
int fbbev_msda_fwd_fused(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* ref_points, const float* offsets, const float* attn_weight, int batch,
                         int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                         int head_stride, int offsets_head_minor, float* out, fbbev_stream_t stream);

/* Replaces ext_module.ms_deform_attn_backward(..., grad_output, grad_value, grad_sampling_loc,
 *   grad_attn_weight, im2col_step=...) -- multi_scale_deformable_attn_function.py:159-169.
 * The three grad outputs are pre-zeroed by the caller (:155-157) and accumulated into. */
int fbbev_msda_bwd(const float* value, const int64_t* spatial_shapes,
                   const int64_t* level_start_index, const float* sampling_loc,
                   const float* attn_weight, const float* grad_output, int batch, int spatial_size,
                   int num_heads, int channels, int num_levels, int num_query, int num_point,
                   float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                   fbbev_stream_t stream);

/* The same backward without floating-point global atomics (bit-reproducible): grad_value is accumulated per (sample, head,
 * BAND of token rows) in LDS as 64-bit fixed point and every token is written exactly once -- grad_value need NOT be
 * pre-zeroed; grad_sampling_loc / grad_attn_weight are accumulated into as above.  Queries are binned by the token rows
 * their samples reach (two small passes over sampling_loc into `ws`), so raster-ordered BEV queries that sample around
 * themselves cost a halo of re-evaluated samples per band; arbitrary sampling stays correct and gets slower.
 * level_hw_host: HOST array of num_levels (h, w) pairs (the band count is a launch dimension).  ws: device scratch of
 * fbbev_msda_bwd_ws_bytes(...) bytes; 0 from that function (channels not in {4,8,10,16,32}, a level wider than a plane,
 * no host shapes) means this entry forwards to fbbev_msda_bwd -- then grad_value MUST be pre-zeroed, so callers zero it
 * whenever ws_bytes() returned 0. */
size_t fbbev_msda_bwd_ws_bytes(int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                               int num_point, const int32_t* level_hw_host);
int fbbev_msda_bwd_ws(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                      const float* sampling_loc, const float* attn_weight, const float* grad_output, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                      float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                      const int32_t* level_hw_host, void* ws, size_t ws_bytes, fbbev_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Fused backward-projection sampling (additive)
 * -------------------------------------------------------------------------------------------- */

/* Replaces bevformer_encoder.point_sampling
 *   -- bevformer_utils/bevformer_encoder.py:91-120 (3 batched inverses + 3 broadcast matmuls).
 * xs (X), ys (Y), zs (Za): voxel-centre axes of get_reference_points '3d' (:66-75); camera tensors as in
 * fbbev_lidar_coor; ogfH/ogfW = data_config['input_size'].  Outputs: ref_cam (N,B,Y*X,Za,2) normalised
 * pixel coordinates, mask (N,B,Y*X,Za) bool, qdepth (N,B,Y*X,Za) camera-frame depth.  ref_cam 8-byte aligned (a point's (u, v) is
 * one store), FBBEV_E_UNSUPPORTED otherwise. */
int fbbev_point_sampling(const float* xs, const float* ys, const float* zs, const float* rots,
                         const float* trans, const float* intrins, const float* post_rots,
                         const float* post_trans, const float* bda, int B, int N, int Y, int X, int Za,
                         float ogfH, float ogfW, float* ref_cam, uint8_t* mask, float* qdepth,
                         fbbev_stream_t stream);

/* Replaces the sampling core of DA_SpatialCrossAttention.forward + DA_MSDeformableAttention.forward
 *   -- bevformer_utils/spatial_cross_attention_depth.py:163-216 and :554-595 (6*B nonzero() syncs,
 *   rebatch/pad/scatter Python loops, the (B*6,L,Za,DC) one-hot, two ms_deform_attn_forward launches).
 * value (B*Ncam,S,M,Dh) = value_proj(camera tokens); pred_depth (B*Ncam,DC,H0,W0) = the depth
 * distribution in its native layout (level-0 shape); ref_cam (Ncam,B,Q,Za,2), mask (Ncam,B,Q,Za) bool,
 * qdepth (Ncam,B,Q,Za) from point_sampling (bevformer_encoder.py:91-120); offsets (B,Q,M,L,P,2) =
 * sampling_offsets(query) raw, attn (B,Q,M,L,P) = softmax(attention_weights(query)), both computed once
 * per BEV query; d0/dstep = dbound[0]/dbound[2].  head_minor: bit 0 -> offsets is laid out (B,Q,L,P,M,2), bit 1 -> attn
 * is (B,Q,L,P,M) -- what the Linear layers emit when their output rows are permuted; the heads of a query then
 * read contiguous bytes per sample (offsets head-minor is the fast path of the FB-OCC shapes); bit 2 -> a token's
 * M*head_stride floats are stored chunk-major, (head_stride/4, M, 4), instead of (M, head_stride): the 8 head lanes of
 * a query read one contiguous M*16-byte piece per load (needs head_stride % 4 == 0; same for grad_value in the
 * backward entries).
 * head_stride: floats between two heads inside a value row (0 = Dh, i.e. value is (B*Ncam,S,M,Dh) dense); a value_proj
 * output padded to a multiple of 4 floats per head (Dh = 10 -> 12) makes every head chunk 16-byte aligned and lets the
 * kernel read a corner with 3 dwordx4 loads; the padding floats are ignored (grad_value of the padding stays 0).
 * slots (B,Q,M*Dh) = sum over hit cameras of the depth-weighted deformable sample / max(#hit,1)
 * (the tensor the reference feeds to output_proj, :216-219). */
int fbbev_da_cross_attn_fwd(const float* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const float* pred_depth,
                            const float* ref_cam, const uint8_t* mask, const float* qdepth,
                            const float* offsets, const float* attn, int B, int Ncam, int S, int M, int Dh,
                            int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor,
                            int head_stride, float* slots, fbbev_stream_t stream);

/* fbbev_da_cross_attn_fwd on 16-bit camera tokens (inference option; fp32 accumulate, every other tensor fp32):
 * value_elem_type 0 = f32 (== fbbev_da_cross_attn_fwd), 1 = bf16, 2 = f16.  16-bit rows must be chunk-major
 * (head_minor bit 2) with EIGHT elements per (chunk, head) piece -- a token is (head_stride/8, M, 8) -- and
 * head_stride a multiple of 8 covering Dh (Dh = 10 -> 16): half the gather bytes of the fp32 rows.  Elements are
 * widened exactly, so the result equals the fp32 entry on the widened tokens bit for bit. */
int fbbev_da_cross_attn_fwd_e(const void* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* pred_depth,
                              const float* ref_cam, const uint8_t* mask, const float* qdepth,
                              const float* offsets, const float* attn, int B, int Ncam, int S, int M, int Dh,
                              int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor,
                              int head_stride, int value_elem_type, float* slots, fbbev_stream_t stream);

/* fbbev_da_cross_attn_fwd on a value buffer that carries ONE EXTRA token: (B*Ncam*S + 1) tokens of M*head_stride floats,
 * the last one all +0.0f.  The pipelined sampler (two samples in flight per lane, no branch between issuing a sample's
 * corner loads and blending the previous one) points padded corners and out-of-image samples at that token instead of
 * skipping their loads; the result is the same sum (a padded corner contributes w * 0, spatial_cross_attention_depth.py:
 * 593-595 via the zero padding of the op).  Taken for chunk-major fp32 rows (head_minor = 1 | 4), Za = 4, P % 4 == 0,
 * Dh in {8, 10} with head_stride = Dh rounded up to 4, L*P <= 36; every other shape runs fbbev_da_cross_attn_fwd on the
 * same buffer.  `offset / size` is evaluated as offset * (1 / size) (one ulp of a sub-pixel offset).
 * head_minor | FBBEV_DA_ATTN_LOGITS: `attn` holds the RAW output of the attention_weights Linear and the kernel applies
 * the softmax over each unit's L*P weights while it stages them (no separate softmax launch, no extra pass over the
 * tensor); FBBEV_E_UNSUPPORTED when the shape does not take the pipelined kernel -- query fbbev_da_cross_attn_fwd_zt_fuses_softmax
 * first. */
#define FBBEV_DA_ATTN_LOGITS 0x10
int fbbev_da_cross_attn_fwd_zt_fuses_softmax(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za,
                                             int head_minor, int head_stride);
int fbbev_da_cross_attn_fwd_zt(const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* pred_depth,
                               const float* ref_cam, const uint8_t* mask, const float* qdepth,
                               const float* offsets, const float* attn, int B, int Ncam, int S, int M, int Dh,
                               int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor,
                               int head_stride, int bev_w, float* slots, fbbev_stream_t stream);
/* bev_w: row length of the BEV grid the Q = bev_h * bev_w queries come from (0 = unknown).  With it (and M = 8) a workgroup
 * owns the 8 heads of an 8 x 4 PATCH of the grid and a wave 4 heads of a 4 x 4 sub-patch instead of 32 / 8 consecutive
 * queries of a row: neighbours in both BEV directions sample nearly the same camera tokens for a given head, so a load
 * instruction touches fewer distinct cache lines.  Layouts and results (bit for bit) do not depend on it. */

/* The whole inference cross-attention sampling in ONE kernel, from the BEV query rows to the slots (SURVEY 8b:
 * `fbbev_da_cross_attn_fused`; replaces spatial_cross_attention_depth.py:533-595 + :136-223, i.e. the sampling_offsets and
 * attention_weights Linears, the softmax, both MSDA launches, the rebatch / scatter loops).  Differences to
 * fbbev_da_cross_attn_fwd_zt:
 *   - `planes`: camera tokens (the value_proj output) as HEAD PLANES, (B*Ncam, M, S, Dh) fp32, no padding -- written by
 *     fbbev_rows_linear_x3_planes (or fbbev_rows_to_head_planes from row-major tokens);
 *   - no offsets / attn tensors: `query` (B*Q rows of E = M*Dh floats, query_row_stride apart) [+ `addend` rows with period
 *     addend_period: the positional encoding] is projected inside the kernel with the split-operand bf16 MFMA arithmetic of
 *     fbbev_rows_linear_x3 (~1e-5 relative) from `offsets_fragments` / `attn_fragments` = fbbev_rows_linear_x3_fragments of
 *     sampling_offsets.weight (M*L*P*2, E) / attention_weights.weight (M*L*P, E) in the MODULE's row order, + fp32 biases;
 *   - a workgroup = the M heads of an 8 x 8 patch of the BEV grid (Q = bev_h x bev_w, bev_w required), a wave = one head.
 * Supported: M = 8, Dh in {8, 10}, P = 8, Za = 4, every level >= 2 tokens wide (`min_level_width`: the caller's host-side
 * value), Ncam <= 32 (a lane's hit flags are a bit mask), S < 2^24 tokens per image (24-bit offset multiplies), LDS budget
 * (fbbev_da_cross_attn_fused_supported); query / addend rows, fragments and offsets_bias 16-byte aligned, planes / slots 8-byte;
 * FBBEV_E_UNSUPPORTED otherwise.  Result: the reference's sum in
 * (level, camera, point) order -- equal to fbbev_da_cross_attn_fwd up to fp32 re-association and the projection arithmetic. */
int fbbev_da_cross_attn_fused_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int bev_w);
int fbbev_da_cross_attn_fused(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                              const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                              const float* query, long long query_row_stride, const float* addend,
                              long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                              const float* offsets_bias, const void* attn_fragments, const float* attn_bias, int B,
                              int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                              int bev_w, int min_level_width, float* slots, fbbev_stream_t stream);
/* fbbev_da_cross_attn_fused on 16-bit head planes (round 5): elem_type 1 = bf16, 2 = fp16 (0 = the entry above), the planes written
 * by fbbev_rows_linear_x3_planes_e.  The reference keeps the camera tokens in fp32; this is the storage option of
 * DA_SpatialCrossAttention.value_dtype on the one-kernel route (half the gather bytes; tokens rounded once, products / sums fp32). */
int fbbev_da_cross_attn_fused_e(const void* planes, int elem_type, const int64_t* spatial_shapes, const int64_t* level_start_index,
                              const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                              const float* query, long long query_row_stride, const float* addend,
                              long long addend_row_stride, long long addend_period, const void* offsets_fragments,
                              const float* offsets_bias, const void* attn_fragments, const float* attn_bias, int B,
                              int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep,
                              int bev_w, int min_level_width, float* slots, fbbev_stream_t stream);
/* fbbev_da_cross_attn_fused followed, inside the same workgroups (all 8 heads of a patch in one 512-thread workgroup), by the block's
 * tail: out = LayerNorm(output_proj(slots) + residual) (spatial_cross_attention_depth.py:223-226 + the layer's norm).  The arguments of
 * fbbev_msda_self_fused_ln; M*Dh % 16 == 0. */
int fbbev_da_cross_attn_fused_ln(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                 const float* query, long long query_row_stride, const float* addend, long long addend_row_stride,
                                 long long addend_period, const void* offsets_fragments, const float* offsets_bias,
                                 const void* attn_fragments, const float* attn_bias, const void* out_fragments, const float* out_bias,
                                 const float* residual, long long residual_row_stride, const float* ln_weight, const float* ln_bias,
                                 float ln_eps, int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int DC, float d0,
                                 float dstep, int bev_w, int min_level_width, float* out, fbbev_stream_t stream);

/* The BEV self-attention of the encoder layer in ONE kernel, from the query rows to the attention output (before output_proj):
 * mmcv MultiScaleDeformableAttention.forward as bevformer_encoder.py:327-341 calls it -- sampling_offsets / attention_weights
 * Linears (split-operand bf16 MFMA inside the kernel, fragments of the MODULE-ordered weights), softmax over the head's points,
 * `loc = ref + offset / (W, H)`, the bilinear samples of ms_deform_attn_forward.  `planes`: the value_proj output as head planes
 * (B, M, S, Dh) with S = level_h * level_w value tokens per sample (fbbev_rows_linear_x3_planes, tokens_per_image = S);
 * reference_points (B, Q, 1, 2) normalised (x, y); query (+ addend rows, period addend_period) as fbbev_da_cross_attn_fused;
 * out (B, Q, M*Dh).  Supported: M = 8, Dh in {8, 10}, ONE level, P = 4 (mmcv's defaults = the FB-OCC configs), Q = a bev_h x
 * bev_w grid, level_w >= 2; FBBEV_E_UNSUPPORTED otherwise (callers fall back to fbbev_msda_fwd_fused). */
int fbbev_msda_self_fused_supported(int B, int S, int M, int Dh, int L, int Q, int P, int bev_w);
int fbbev_msda_self_fused(const float* planes, const float* reference_points, const float* query, long long query_row_stride,
                          const float* addend, long long addend_row_stride, long long addend_period,
                          const void* offsets_fragments, const float* offsets_bias, const void* attn_fragments,
                          const float* attn_bias, int B, int S, int M, int Dh, int L, int Q, int P, int bev_w, int level_h,
                          int level_w, float* out, fbbev_stream_t stream);
/* fbbev_msda_self_fused followed, inside the same workgroups, by the attention block's tail: out = LayerNorm(output_proj(attention)
 * + residual) (bevformer_encoder.py:250-377, operation_order (self_attn, norm, ...)).  out_fragments: output_proj.weight as split bf16
 * fragments (fbbev_rows_linear_x3_fragments); residual rows (B*Q, residual_row_stride) or NULL.  Same arithmetic as
 * fbbev_rows_linear_x3_ln on the attention output (three-MFMA split operands, two-pass LayerNorm statistics); M*Dh % 16 == 0. */
int fbbev_msda_self_fused_ln(const float* planes, const float* reference_points, const float* query, long long query_row_stride,
                             const float* addend, long long addend_row_stride, long long addend_period,
                             const void* offsets_fragments, const float* offsets_bias, const void* attn_fragments,
                             const float* attn_bias, const void* out_fragments, const float* out_bias, const float* residual,
                             long long residual_row_stride, const float* ln_weight, const float* ln_bias, float ln_eps, int B, int S,
                             int M, int Dh, int L, int Q, int P, int bev_w, int level_h, int level_w, float* out,
                             fbbev_stream_t stream);

/* row-major camera tokens (n_rows = B*Ncam*S rows of M*Dh floats, module order (head, channel)) -> head planes (B*Ncam, M, S, Dh) */
int fbbev_rows_to_head_planes(const float* rows, long long n_rows, int tokens_per_image, int M, int Dh, float* planes,
                              fbbev_stream_t stream);

/* Backward of fbbev_da_cross_attn_fwd in one launch -- replaces the autograd chain of the reference's training step
 * through DA_SpatialCrossAttention / DA_MSDeformableAttention (two MultiScaleDeformableAttnFunction backward launches,
 * multi_scale_deformable_attn_function.py:137-172, plus the rebatch / one-hot / scatter index ops and their host syncs).
 * Inputs as the forward + grad_slots (B,Q,M*Dh).  Outputs, all PRE-ZEROED by the caller (the mmcv convention,
 * :159-169): grad_value like value and grad_pred_depth like pred_depth (accumulated with fp32 atomics), grad_offsets /
 * grad_attn in the layouts of offsets / attn (head_minor bits).  Dh <= 32, else FBBEV_E_UNSUPPORTED. */
int fbbev_da_cross_attn_bwd(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                            const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                            const float* offsets, const float* attn, const float* grad_slots, int B, int Ncam, int S,
                            int M, int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor,
                            int head_stride, float* grad_value, float* grad_pred_depth, float* grad_offsets,
                            float* grad_attn, fbbev_stream_t stream);

/* fbbev_da_cross_attn_bwd with the value gradient accumulated in LDS planes instead of global atomics, three launches:
 * (A) the unit-owned gradients (attention weights, sampling offsets, depth distribution) by a kernel with the forward's
 * lane mapping; (B) the value gradient: a workgroup owns (sample, head, chunk of BEV queries), keeps the head's gradient
 * plane of one camera at a time in LDS as 64-bit fixed point (bit-reproducible) and writes it to its slice of `ws` -- one
 * launch per TOKEN REGION when the pyramid's plane does not fit LDS (whole levels, or row bands of a large level);
 * (C) the slices are summed into grad_value in fixed order (written, not accumulated -- no pre-zeroing needed for it;
 * grad_pred_depth / grad_offsets / grad_attn are accumulated as in fbbev_da_cross_attn_bwd and must be pre-zeroed).
 * level_hw_host: HOST array of num_levels (h, w) pairs -- what spatial_shapes holds on the device -- so that the regions
 * are planned without a device read; NULL plans only the case where the whole pyramid fits one plane (<= 717 tokens at
 * head_stride 12).  ws: fbbev_da_cross_attn_bwd_ws_bytes(...) bytes, 16-byte aligned; 0 bytes = the plan rejects the shape
 * (head dims other than 4 / 8 / 10 / 16, more than 8 points per level, head_stride not a multiple of 4 or > 16).  With
 * ws == NULL, too small, or a rejected shape the call IS fbbev_da_cross_attn_bwd (same results up to the order of the
 * fp32 adds).
 * Round 4, OUTPUT-OWNED planes: when the launch has at least one workgroup per CU (token regions x B x Ncam x M >= 256;
 * FBBEV_DA_BWD_OWNED=1 / 0 forces / forbids it) step (B) is instead: per-(sample, camera) lists of hit RECORDS (query, camera count,
 * depth weights, reference points: 64 bytes, in list order) in `ws` (k_da_bwd_hitlist), then ONE launch in which a workgroup owns the plane of one (sample, camera, head,
 * token region), walks the camera's hit list and writes its tokens of grad_value directly -- no partial planes, no step
 * (C); `ws` shrinks from the partial planes (551 MB at the configs[2] pyramid) to the hit records (64 bytes per (camera, query): 61 MB).  The fixed-point
 * scale is then the SAMPLE's max |grad_slots| (folded by kernel (A); round 5: per sample, round 4: per call): every addend
 * of a sample is quantised to 2^-30 of that maximum, i.e. contributions below ~1e-9 of the sample's largest upstream gradient
 * round to zero (an outlier -- AMP loss scaling -- costs resolution in its own sample only); a non-finite upstream gradient
 * makes that sample's grad_value NaN.  Same bits run to run; configs[2] pyramid, B = 4: 2.39 -> 1.18 ms. */
size_t fbbev_da_cross_attn_bwd_ws_bytes(int B, int Ncam, int S, int M, int Dh, int Q, int head_stride,
                                        int num_levels, int num_points, const int32_t* level_hw_host);
/* fbbev_da_cross_attn_bwd_ws_bytes covers BOTH LDS-plane routes (the launch picks one from arguments the query does not see: the
 * number of Z anchors, pointer alignment), i.e. the maximum of their sizes.  With the anchor count the launch will pass, this
 * query returns the size of the route that launch takes (61 MB instead of 551 MB at the configs[2] pyramid). */
size_t fbbev_da_cross_attn_bwd_ws_bytes_za(int B, int Ncam, int S, int M, int Dh, int Q, int head_stride, int num_levels,
                                           int num_points, int num_z_anchors, const int32_t* level_hw_host);
int fbbev_da_cross_attn_bwd_ws(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                               const float* qdepth, const float* offsets, const float* attn,
                               const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                               int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                               float* grad_value, float* grad_pred_depth, float* grad_offsets,
                               float* grad_attn, const int32_t* level_hw_host, void* ws, size_t ws_bytes,
                               fbbev_stream_t stream);
/* out[bc][i] = (sum over z of volume[bc][z][i]) / divisor for a materialised (B*C, Z, Y*X) fp32 volume: the training path's
 * `bev_feat.mean(-1)` (fbocc.py:359; divisor = Z) and the Z-sum that the backward of the re-add `refined[..., None] + bev_feat`
 * (fbocc.py:365-366) sends to the refined BEV (divisor = 1), one HBM-bound pass each.  YX % 4 == 0, 16-byte aligned. */
int fbbev_volume_zreduce(const float* volume, long long n_bc, int Z, long long YX, float divisor, float* out, fbbev_stream_t stream);

/* The same for a Z-INNERMOST volume (B*C*Y*X pillars of Z contiguous floats): an upstream gradient that arrives contiguous in the
 * module's (B,C,Y,X,Z) output shape; and its re-layout (B*C, Y*X, Z) -> (B*C, Z, Y*X) for the pooling backward (the `.contiguous()`
 * autograd would otherwise run as a strided ATen copy).  Z % 4 == 0, 16-byte aligned, src != dst. */
int fbbev_volume_zreduce_inner(const float* volume, long long n_pillars, int Z, float divisor, float* out, fbbev_stream_t stream);
int fbbev_volume_z_to_front(const float* src, long long n_bc, int Z, long long YX, float* dst, fbbev_stream_t stream);

/* The TRAINING forward of the DA cross-attention on head planes (round 4; k_da_fwd_planes): what fbbev_da_cross_attn_fwd computes
 * (spatial_cross_attention_depth.py:136-223, 513-595 with projected offsets and softmaxed weights handed in, as autograd owns the
 * two Linears), with the camera tokens as (B*Ncam, M, S, Dh) planes -- fbbev_value_rows_to_head_planes re-lays the value rows
 * (plain, or chunk-major when `interleaved`) out -- and the one-kernel sampler's mapping: a workgroup = the 8 heads of an 8 x 8
 * patch of a bev_w-wide query grid (bev_w = 0: 64 consecutive queries), a wave = one head.  offsets / attn in the layouts of
 * fbbev_da_cross_attn_fwd (head_minor bits 0 / 1).  M = 8, Dh in {8, 10}, 8 points, 4 anchors, every level >= 2 tokens wide
 * (min_level_width: the caller's host-side knowledge); otherwise FBBEV_E_UNSUPPORTED.  slots (B, Q, M*Dh) is written. */
int fbbev_value_rows_to_head_planes(const float* value, long long n_tokens, int S, int M, int Dh, int head_stride, int interleaved,
                                    float* planes, fbbev_stream_t stream);
int fbbev_da_cross_attn_fwd_planes_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za);
int fbbev_da_cross_attn_fwd_planes(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                   const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                   const float* offsets, const float* attn, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                                   int P, int Za, int DC, float d0, float dstep, int head_minor, int bev_w, int min_level_width,
                                   float* slots, fbbev_stream_t stream);

/* fbbev_da_cross_attn_bwd_ws for queries that form a (Q / bev_w) x bev_w BEV grid (bevformer_encoder.py:91-120: the reference's
 * queries always do).  Round 4: on the output-owned route the unit gradients (step A) run on HEAD PLANES with the forward's mapping
 * (k_da_bwd_unit_planes: the camera tokens are re-laid out as (B*Ncam, M, S, Dh) planes in `ws`, a workgroup = the 8 heads of an
 * 8 x 8 patch of queries, a wave = one head) when M = 8, Dh in {8, 10}, 8 points, 4 anchors, levels >= 2 tokens wide
 * (FBBEV_DA_BWD_UNIT_PLANES=0 keeps k_da_cross_attn_bwd_unit).  bev_w = 0 (or not a divisor of Q): patches of 64 consecutive
 * queries -- what fbbev_da_cross_attn_bwd_ws does. */
int fbbev_da_cross_attn_bwd_ws_grid(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* pred_depth, const float* ref_cam, const uint8_t* mask,
                               const float* qdepth, const float* offsets, const float* attn,
                               const float* grad_slots, int B, int Ncam, int S, int M, int Dh, int L, int Q,
                               int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                               float* grad_value, float* grad_pred_depth, float* grad_offsets,
                               float* grad_attn, const int32_t* level_hw_host, void* ws, size_t ws_bytes, int bev_w,
                               fbbev_stream_t stream);

/* Backward of fbbev_da_cross_attn_fused for a training step that ran the one-kernel forward (round 6): the camera tokens arrive as
 * the HEAD PLANES (B*Ncam, M, S, Dh) the forward sampled (fbbev_rows_linear_x3_planes), offsets (B,Q,L,P,M,2) / attn as for
 * fbbev_da_cross_attn_bwd_ws_grid (head_minor bits 0 / 1), grad_value in the row layout of that entry (head_stride, head_minor bit 2).
 * On this route grad_value, grad_offsets and grad_attn are WRITTEN in full (no pre-zeroing); grad_pred_depth is accumulated into with
 * fp32 atomics and must be zeroed by the caller.  Workspace: fbbev_da_cross_attn_bwd_ws_bytes_za.  Applies when
 * fbbev_da_cross_attn_bwd_planes_supported returns 1 (the output-owned plane route with unit gradients on head planes: M = 8,
 * Dh in {8, 10}, 8 points, 4 anchors, levels >= 2 tokens wide, >= 256 planes, queries on a bev_w-wide grid); else FBBEV_E_UNSUPPORTED.
 * Reference: autograd through spatial_cross_attention_depth.py:136-223,513-595 and mmcv ms_deform_attn_backward. */
int fbbev_da_cross_attn_bwd_planes_supported(int B, int Ncam, int S, int M, int Dh, int L, int Q, int P, int Za, int head_stride,
                                             const int32_t* level_hw_host, int bev_w);
int fbbev_da_cross_attn_bwd_planes(const float* planes, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                   const float* pred_depth, const float* ref_cam, const uint8_t* mask, const float* qdepth,
                                   const float* offsets, const float* attn, const float* grad_slots, int B, int Ncam, int S, int M,
                                   int Dh, int L, int Q, int P, int Za, int DC, float d0, float dstep, int head_minor, int head_stride,
                                   float* grad_value, float* grad_pred_depth, float* grad_offsets, float* grad_attn,
                                   const int32_t* level_hw_host, void* ws, size_t ws_bytes, int bev_w, fbbev_stream_t stream);

/* Training backward of the fused lift-splat:  replaces QuickCumsumCuda.backward (bev_pool.py:39-78 --
 * argsort of ranks_feat, mask-built intervals [2 host syncs], the permute().contiguous() of the gradient)
 * and bev_pool_v2_backward / bev_pool_v2_grad_kernel (src/bev_pool_cuda.cu:52-100,128-135).
 * No host sync: the index tensors and counts are the device-side outputs of fbbev_rank_build /
 * fbbev_lift_rank_build of the forward pass (padded arrays, valid prefixes counts[0]=P, counts[1]=I).
 * out_grad is the gradient of the (B,C,Z,Y,X) output in THAT layout (element strides og_stride_b /
 * og_stride_c, 0 = contiguous; the (Z,Y,X) block of a channel contiguous) -- no channels-last copy.
 * depth (B,N,D,H,W), feat (B,N,H,W,C) as in the forward; depth_grad / feat_grad have the same shapes and
 * are written completely (zeros for dropped points / untouched pixels): they need not be pre-zeroed.
 * feat_grad: in-order fmaf chain over the kept depth bins of each pixel in ascending d (the reference's
 * order is that of its unstable argsort); depth_grad: C-long dot product reduced across lanes.
 * Requires C % 4 == 0, C <= 256 (C % 8 == 0 above 128), (Y*X) % 4 == 0, 16-byte aligned pointers. */
size_t fbbev_pool_dense_bwd_workspace_bytes(int B, int N, int D, int H, int W, int C, int Z, int Y, int X);
int fbbev_bev_pool_v2_dense_bwd(const float* out_grad, long long og_stride_b, long long og_stride_c,
                                const float* depth, const float* feat, const int32_t* ranks_depth,
                                const int32_t* interval_rank, const int32_t* interval_starts,
                                const int32_t* counts, int n_intervals_max, int B, int N, int D, int H, int W,
                                int C, int Z, int Y, int X, float* depth_grad, float* feat_grad,
                                void* workspace, size_t workspace_bytes, fbbev_stream_t stream);
/* fbbev_bev_pool_v2_dense_bwd with a second upstream gradient zgrad (B, C, Y, X) that EVERY z plane receives, scaled:
 * out_grad_eff[b,c,z,y,x] = out_grad[b,c,z,y,x] + zscale * zgrad[b,c,y,x] -- the backward of `bev_feat.mean(-1)` (fbocc.py:359:
 * zscale = 1 / Z) folded into the gradient read instead of an expand + add over the whole volume. */
int fbbev_bev_pool_v2_dense_bwd_z(const float* out_grad, long long og_stride_b, long long og_stride_c, const float* zgrad,
                                  float zscale, const float* depth, const float* feat, const int32_t* ranks_depth,
                                  const int32_t* interval_rank, const int32_t* interval_starts, const int32_t* counts,
                                  int n_intervals_max, int B, int N, int D, int H, int W, int C, int Z, int Y, int X,
                                  float* depth_grad, float* feat_grad, void* workspace, size_t workspace_bytes,
                                  fbbev_stream_t stream);


/* Temporal history alignment -- replaces FBOCC.generate_grid + the 5-D F.grid_sample of FBOCC.fuse_history
 *   -- mmdet3d/models/fbbev/detectors/fbocc.py:169-205 and :264-275.
 * fbbev_history_flow: rt_flow[b] (4x4, row-major) = inv(feat2bev) . history_forward_augs[b] . curr_to_prev_ego_rt[b]
 *   . inv(forward_augs[b]) . feat2bev (:184-203), forward_augs = the homogeneous embedding of bda[b] (:36-41),
 *   feat2bev = diag(dx3) with translation lower3 = bx - dx/2 (:184-195).  dx3 / lower3 are HOST pointers (x,y,z).
 * fbbev_history_warp: out[b,ch,z,y,x] = trilinear sample (align_corners=True, zero padding) of history[b,ch] at
 *   rt_flow[b] . (x,y,z,1) -- the sampling grid is never materialised.  history / out: (B,CH,Z,Y,X) f32 with batch
 *   strides in elements (0 = contiguous), so `out` may be a channel slice of the next history buffer.
 *   Z, Y, X >= 2 (the reference normalises by size-1). */
int fbbev_history_flow(const float* history_forward_augs, const float* curr_to_prev_ego_rt, const float* bda,
                       const float* dx3, const float* lower3, int B, float* rt_flow, fbbev_stream_t stream);
int fbbev_history_warp(const float* history, long long history_stride_b, const float* rt_flow, int B, int CH, int Z,
                       int Y, int X, float* out, long long out_stride_b, fbbev_stream_t stream);
/* The same with a storage element type for history AND out: elem_type 0 = f32, 1 = bf16, 2 = f16 (BASELINE configs[4]
 * names fp16: the 16-frame history of a 400x400x16 grid is 13 GB per sample in fp32).  The taps are widened exactly,
 * the trilinear sum is the same fp32 fmaf chain, the result is rounded ONCE (nearest-even) at the store; strides are in
 * elements.  fbbev_history_conv_e reads a frame buffer of that element type (fp32 MFMA, fp32 output, as before). */
#define FBBEV_ELEM_F32 0
#define FBBEV_ELEM_BF16 1
#define FBBEV_ELEM_F16 2
int fbbev_history_warp_e(const void* history, long long history_stride_b, const float* rt_flow, int B, int CH, int Z,
                         int Y, int X, void* out, long long out_stride_b, int elem_type, fbbev_stream_t stream);

/* LayerNorm over the last dimension of (rows, C) f32, optional residual:  out = LN(x + residual) * weight + bias
 * (biased variance, eps inside the square root: torch.nn.LayerNorm = mmcv build_norm_layer('LN'), the `norm` steps of
 * BEVFormerEncoderLayer, bevformer_encoder.py:250-377).  C % 4 == 0, C <= 128, 16-byte aligned pointers, else
 * FBBEV_E_UNSUPPORTED (callers keep torch's kernel).  residual may be NULL; out may alias x. */
int fbbev_layernorm(const float* x, const float* residual, const float* weight, const float* bias, float eps,
                    long long rows, int C, float* out, fbbev_stream_t stream);

/* Backward of fbbev_layernorm without a residual (training of the backward projection: the `norm` steps of
 * BEVFormerEncoderLayer under autograd, bevformer_encoder.py:250-377; replaces ATen's native_layer_norm_backward):
 *   grad_x[row] = inv * (g - mean(g) - xhat * mean(g * xhat)),   g = grad_out[row] * weight, xhat = (x[row] - mean) * inv
 * and per workgroup w one pair of partial parameter gradients  partial[w][0][C] = sum grad_out * xhat (-> weight),
 * partial[w][1][C] = sum grad_out (-> bias) over the rows that workgroup walked; the caller sums the
 * fbbev_layernorm_bwd_partials(rows) partial rows (fixed order: deterministic, no atomics).  mean / inv are recomputed
 * from x.  Same shape limits and return codes as fbbev_layernorm. */
int fbbev_layernorm_bwd_partials(long long rows);
int fbbev_layernorm_bwd(const float* x, const float* grad_out, const float* weight, float eps, long long rows, int C,
                        float* grad_x, float* partial, fbbev_stream_t stream);

/* Weight and bias gradient of a row-wise linear layer y = x W^T + b (training of the backward projection: autograd's
 * `grad_out.t().mm(x)` / `grad_out.sum(0)` behind every nn.Linear of bevformer_encoder.py:206-377 and
 * spatial_cross_attention_depth.py:432-436,464 -- vendor fp32 GEMMs + ATen reductions in the reference):
 *   grad_weight (out_features, in_features) = grad_out^T x,   grad_bias (out_features) = column sums of grad_out (NULL: skipped)
 * for grad_out (rows, out_features) and x (rows, in_features), row strides in floats (0 = dense); x_addend (period, in_features),
 * optional: the layer's input rows were x[r] + x_addend[r % period] (query + query_pos, never materialised).  Arithmetic: the split-operand bf16
 * MFMA of fbbev_rows_linear_x3 (~1e-5 relative), rows split over workgroups, partial results summed in a FIXED order (bit-identical
 * run to run, no atomics) through `workspace` (fbbev_rows_wgrad_x3_ws_bytes bytes, 16-byte aligned).  in_features % 4 == 0,
 * out_features % 4 == 0, strides % 4 == 0, 16-byte aligned pointers, else FBBEV_E_UNSUPPORTED (ws_bytes returns 0). */
size_t fbbev_rows_wgrad_x3_ws_bytes(long long rows, int in_features, int out_features);
int fbbev_rows_wgrad_x3(const float* grad_out, long long ld_grad, const float* x, long long ldx, const float* x_addend,
                        long long addend_row_stride, long long addend_period, long long rows, int in_features, int out_features,
                        float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes, fbbev_stream_t stream);

/* Training epilogue of fbbev_rows_linear_x3 (same arithmetic, shapes and return codes):
 *   out = ((x [+ addend[r % period]]) W^T + bias) [ReLU]) * [mask > 0] + residual
 * mask (rows, out_features), optional: the saved output of a ReLU whose backward this product is (ATen threshold_backward folded
 * into the dgrad's store); residual (rows, out_features), optional, MAY be `out` itself (a running sum of gradients): the forward
 * recomputations and dgrads of the encoder layer's backward (bevformer_encoder.py:250-377 under autograd) without their
 * neighbouring element-wise passes. */
int fbbev_rows_linear_x3_train(const float* x, long long x_row_stride, const float* addend, long long addend_row_stride,
                               long long addend_period, const void* fragments, const float* bias, long long rows, int in_features,
                               int out_features, int relu, const float* residual, long long residual_row_stride, const float* mask,
                               long long mask_row_stride, float* out, long long out_row_stride, fbbev_stream_t stream);

/* out (N) = sum over b of x (B, N) [+ x2 (B, N)], ascending b, N % 4 == 0: the batch sum behind a parameter every sample shares
 * (autograd's sum-to-size of `query + query_pos`, backward_projection.py:96-99 `lss_bev + bev_embedding`). */
int fbbev_sum_leading(const float* x, const float* x2, int B, long long N, float* out, fbbev_stream_t stream);
/* softmax over groups of `group` consecutive floats (4, 8, 16 or 32: the num_levels * num_points attention logits of one (query,
 * head), spatial_cross_attention_depth.py:541-546 / mmcv MultiScaleDeformableAttention) and its backward
 * grad_x = y * (grad_y - sum(y * grad_y)); grad_x may be grad_y.  16-byte aligned pointers, else FBBEV_E_UNSUPPORTED. */
int fbbev_softmax_groups(const float* x, long long n_groups, int group, float* y, fbbev_stream_t stream);
int fbbev_softmax_groups_bwd(const float* y, const float* grad_y, long long n_groups, int group, float* grad_x, fbbev_stream_t stream);

/* Read-ahead of a kernel's gather sources: reads up to 8 spans (pointer, bytes) once with 16-byte loads and discards the data, so that
 * the lines are back in the memory-side cache when a later kernel's dependent gathers start (the dense pooling at the end of the
 * forward-backward step: its index tensors, depth and feature rows were written ~0.7 GB of intermediate traffic earlier).  Values
 * are never used; no effect on results. */
int fbbev_touch(const void* const* spans, const size_t* bytes, int n, fbbev_stream_t stream);

/* Diagnostics: fill n floats (n % 4 == 0) with 16-byte stores of one cache policy (0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 nt,
 * 6 sc0 sc1 nt, 7 sc0): measures what a kernel's stores leave behind for the next kernel's store stream (tools/dbg_store_policy.py). */
int fbbev_diag_fill(float* p, long long n, int policy, fbbev_stream_t stream);
/* out (len) = sum over the n rows of part (n, len) in a fixed association: the per-workgroup partial parameter gradients of
 * fbbev_layernorm_bwd (n = fbbev_layernorm_bwd_partials(rows), len = 2 C). */
int fbbev_sum_partials(const float* part, int n, long long len, float* out, fbbev_stream_t stream);

/* Row-wise linear layer  out[r, :] = x[r, :] . W^T + bias (+ ReLU)  for the (B*Q, C) query rows of the backward projection
 * (inference): replaces the F.linear calls of spatial_cross_attention_depth.py:533-540 (sampling_offsets, attention_weights),
 * :494-500 (value_proj, output_proj) and the FFN of the encoder layer (mmcv FFN: Linear + ReLU + Linear) -- vendor fp32 GEMMs in
 * the reference.  Arithmetic: bf16 MFMA with SPLIT operands (v = hi + lo, three MFMAs per product, fp32 accumulation): ~1e-5
 * relative to the fp32 result.  Two steps: fbbev_rows_linear_x3_fragments turns W (out_features, in_features) row-major fp32
 * into split MFMA fragments (once per weight version; fbbev_rows_linear_x3_fragment_bytes bytes, 16-byte aligned), then
 * fbbev_rows_linear_x3 per call.  Row strides in floats (0 = dense).  in_features % 8 == 0, out_features % 4 == 0, strides % 4 == 0,
 * 16-byte aligned pointers, else FBBEV_E_UNSUPPORTED (callers keep the vendor GEMM).  bias may be NULL. */
size_t fbbev_rows_linear_x3_fragment_bytes(int in_features, int out_features);
int fbbev_rows_linear_x3_fragments(const float* weight, int in_features, int out_features, void* fragments, size_t fragment_bytes,
                                   fbbev_stream_t stream);
int fbbev_rows_linear_x3(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                         int in_features, int out_features, int relu, float* out, long long out_row_stride, fbbev_stream_t stream);
/* The same with  x[r, :] + addend[r % addend_period, :]  as the row (one fp32 add per element before the product): the
 * `query = query + query_pos` pass of the attention modules (spatial_cross_attention_depth.py:122-123, mmcv
 * MultiScaleDeformableAttention.forward) folded into the projections that consume the sum.  addend: (addend_period,
 * in_features) rows, stride in floats (0 = dense). */
int fbbev_rows_linear_x3_add(const float* x, long long x_row_stride, const float* addend, long long addend_row_stride,
                             long long addend_period, const void* fragments, const float* bias, long long rows, int in_features,
                             int out_features, int relu, float* out, long long out_row_stride, fbbev_stream_t stream);
/* out = LayerNorm(x W^T + b [+ residual]) -- torch.nn.LayerNorm over the out_features outputs of a row (two-pass statistics,
 * biased variance, eps inside the square root) in the store epilogue of fbbev_rows_linear_x3: the `output_proj -> + residual ->
 * norm` tail of the encoder layer's attention blocks and FFN (bevformer_encoder.py:250-377) without a separate LayerNorm pass.
 * out_features <= 128; residual rows optional (stride in floats, 0 = dense). */
int fbbev_rows_linear_x3_ln(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                            int in_features, int out_features, const float* residual, long long residual_row_stride,
                            const float* ln_weight, const float* ln_bias, float ln_eps, float* out, long long out_row_stride,
                            fbbev_stream_t stream);
/* The FFN pair of the encoder layer in ONE kernel: out = [LayerNorm](W2 relu(W1 x + b1) + b2 [+ residual]) -- mmcv FFN as the encoder
 * layer configures it (Linear + ReLU, Linear, add_identity; bevformer_encoder.py:250-377) and, with ln_weight, the layer's following
 * LayerNorm.  The hidden rows never leave the CU (as two launches they are written and re-read: 205 MB each way at 160 000 rows).
 * Fragments: fbbev_rows_linear_x3_fragments of W1 (hidden, in_features) / W2 (out_features, hidden); the split-operand arithmetic
 * of fbbev_rows_linear_x3 in both GEMMs.  in_features <= 96, hidden % 64 == 0, out_features <= 80. */
int fbbev_rows_ffn_x3(const float* x, long long x_row_stride, const void* w1_fragments, const float* b1, const void* w2_fragments,
                      const float* b2, long long rows, int in_features, int hidden, int out_features, const float* residual,
                      long long residual_row_stride, const float* ln_weight, const float* ln_bias, float ln_eps, float* out,
                      long long out_row_stride, fbbev_stream_t stream);
/* The cross-attention block's tail AND the FFN block of the encoder layer in ONE kernel (round 5) -- bevformer_encoder.py:250-377 with
 * operation_order (..., 'cross_attn', 'norm', 'ffn', 'norm'):
 *   y1  = LayerNorm0(x W0^T + b0 + residual0)        DA_SpatialCrossAttention's output_proj + `+ inp_residual`
 *                                                    (spatial_cross_attention_depth.py:222-223) + the layer's norm
 *   out = LayerNorm1(y1 + W2 relu(W1 y1 + b1) + b2)  mmcv FFN (add_identity) + the layer's last norm
 * x (rows, embed) = the attention slots, residual0 = the block's input rows (optional).  y1 stays in the wave's registers (it is the
 * FFN's residual) and is re-laid out through LDS into GEMM 1's operand fragments: replaces fbbev_rows_linear_x3_ln +
 * fbbev_rows_ffn_x3 and the (rows, embed) tensor between them.  Fragments of W0 (embed, embed), W1 (hidden, embed), W2 (embed,
 * hidden) from fbbev_rows_linear_x3_fragments.  embed % 16 == 0, embed <= 80, hidden % 64 == 0. */
int fbbev_rows_tail_ffn_x3(const float* x, long long x_row_stride, const void* w0_fragments, const float* b0,
                           const float* residual0, long long residual0_row_stride, const float* ln0_weight, const float* ln0_bias,
                           float ln0_eps, const void* w1_fragments, const float* b1, const void* w2_fragments, const float* b2,
                           long long rows, int embed, int hidden, const float* ln1_weight, const float* ln1_bias, float ln1_eps,
                           float* out, long long out_row_stride, fbbev_stream_t stream);
/* The same with the result written as PLANES: rows = images x tokens_per_image, out (images, embed, tokens_per_image) -- the
 * (B, C, Y, X) tensor BackwardProjection returns (backward_projection.py:129: permute + view + contiguous of the encoder's rows),
 * stored by the layer's last kernel instead of rows + a transposing pass; same values.  rows % tokens_per_image == 0. */
int fbbev_rows_tail_ffn_x3_planes(const float* x, long long x_row_stride, const void* w0_fragments, const float* b0,
                                  const float* residual0, long long residual0_row_stride, const float* ln0_weight,
                                  const float* ln0_bias, float ln0_eps, const void* w1_fragments, const float* b1,
                                  const void* w2_fragments, const float* b2, long long rows, int embed, int hidden,
                                  const float* ln1_weight, const float* ln1_bias, float ln1_eps, long long tokens_per_image,
                                  float* out, fbbev_stream_t stream);
/* fbbev_rows_linear_x3 with the result written as HEAD PLANES: rows = (B*Ncam) x tokens_per_image camera tokens, out_features =
 * M * head_dim in the module's (head, channel) order, out (B*Ncam, M, tokens_per_image, head_dim) -- the value_proj of the
 * cross-attention feeding fbbev_da_cross_attn_fused (spatial_cross_attention_depth.py:522-530).  head_dim even, <= 65535 and
 * out_features <= 65535 (the epilogue's index arithmetic), else FBBEV_E_UNSUPPORTED. */
int fbbev_rows_linear_x3_planes(const float* x, long long x_row_stride, const void* fragments, const float* bias,
                                long long rows, int in_features, int out_features, int tokens_per_image, int head_dim,
                                float* out, fbbev_stream_t stream);
/* The same projection with the planes stored in 16 bits (elem_type 1 bf16, 2 fp16, 0 = fp32): one nearest-even rounding of the fp32 result. */
int fbbev_rows_linear_x3_planes_e(const float* x, long long x_row_stride, const void* fragments, const float* bias, long long rows,
                                  int in_features, int out_features, int tokens_per_image, int head_dim, int elem_type, void* out,
                                  fbbev_stream_t stream);

/* The two 1x1x1 convolutions of the temporal fusion in one fp32-MFMA kernel (inference): replaces
 * history_keyframe_time_conv + history_keyframe_cat_conv of FBOCC.fuse_history (fbocc.py:111-127, 289-310) once the
 * eval-mode batch norms are folded into the weights and the time channel into a per-frame bias:
 *   out[b,:,n] = relu( bias2 + sum_t w2[:, t*C:(t+1)*C] . relu( w1 . feats[b, t*C:(t+1)*C, n] + bias1[b*T1+t] ) )
 * feats (B, T1*C, N) f32 with batch stride feats_stride_b (elements, 0 = contiguous), w1 (C,C), bias1 (B*T1,C),
 * w2 (Cout, T1*C), bias2 (Cout), out (B,Cout,N).  C, Cout multiples of 16 and <= 128, else FBBEV_E_UNSUPPORTED.
 * workspace (optional, may be NULL): (1 + T1) * C * max(C, Cout) * 4 bytes of device scratch for fragment-ordered
 * weight copies; with it the C = Cout = 80 shape of FB-OCC takes the register-resident kernel.
 * Exact fp32 arithmetic (v_mfma_f32_16x16x4_f32: k-ordered fmaf chains). */
int fbbev_history_conv(const float* feats, long long feats_stride_b, const float* w1, const float* bias1,
                       const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N, float* out,
                       void* workspace, size_t workspace_bytes, fbbev_stream_t stream);
int fbbev_history_conv_e(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                         const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N, float* out,
                         void* workspace, size_t workspace_bytes, int elem_type, fbbev_stream_t stream);

/* fbbev_history_conv_e with both GEMMs on the bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the folded
 * weights, the frames (exact for a bf16 ring) and the ReLU'd intermediate are rounded to bf16; biases, accumulators
 * and `out` stay fp32.  Opt-in reduced precision -- the reference pins these convolutions to fp32 (fbocc.py:279-282) while
 * BASELINE configs[4] names fp16 for the path; at 400x400x16 the fp32-MFMA kernel is compute bound.  C = Cout in {16, 80};
 * workspace >= (1 + T1) * C * 96 * 2 bytes (bf16 weight fragments), 16-byte aligned.  voxel_major = 1: feats is
 * (B, T1, N, C) (below) instead of (B, T1*C, N). */
int fbbev_history_conv_bf16(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                            const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N, float* out,
                            void* workspace, size_t workspace_bytes, int voxel_major, int elem_type, fbbev_stream_t stream);

/* The same two convolutions at fp32-GRADE precision on the 16-bit MFMAs, operands split into two 16-bit terms (hi + lo), fp32
 * accumulation -- ~1e-5 of the output peak against the fp32 convolutions (the reference's own convolutions run TF32 by default
 * on its hardware).  Convolution 1 takes the ring element as stored (f16 MFMA for an fp16 ring, bf16 MFMA for a bf16 ring) with
 * the weight split in two terms of that type: two MFMAs per product; convolution 2 splits the fp32 intermediate and its weights
 * into two bf16 terms each: three MFMAs per product.  Voxel-major 16-bit ring only (feats (B, T1, N, C), elem_type
 * FBBEV_ELEM_BF16 / FBBEV_ELEM_F16), C = Cout in {16, 80};
 * workspace >= (2 + 2 T1) * (C/16) * ceil(C/32) * 512 * 2 + B * T1 * C * 4 bytes, 16-byte aligned. */
int fbbev_history_conv_bf16x3(const void* feats, long long feats_stride_b, const float* w1, const float* bias1,
                              const float* w2, const float* bias2, int B, int T1, int C, int Cout, int N, float* out,
                              void* workspace, size_t workspace_bytes, int elem_type, fbbev_stream_t stream);

/* One history step on a 16-bit voxel-major ring (the reference's fuse_history, fbocc.py:264-319, after the current frame was put
 * into next[:, 0] by fbbev_history_frame_vm): fbbev_history_warp_vm of history (B, T, N, C) into next[:, 1:] and
 * fbbev_history_conv_bf16x3 of next (B, T+1, N, C) into out (B, Cout, N) -- as a pipeline over `chunks` bands of grid rows (0: the
 * default, 10; 1: the two kernels back to back): the warp of a band runs on `stream`, the convolutions of the band before it
 * on a second stream of the device that the library keeps, joined back into `stream` before the call returns control of it.
 * The gather kernel is bound by memory, the convolutions by the MFMA: side by side each uses what the other leaves idle.  Same
 * kernels, operands and result bits as the two calls.  Shapes, element types and workspace: as fbbev_history_conv_bf16x3.  The second
 * stream and its events are per DEVICE: one call at a time per device.  Measured NOT faster than the two calls. */
int fbbev_history_step_x3_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                             const float* rt_flow, const float* w1, const float* bias1, const float* w2, const float* bias2,
                             int B, int T, int C, int Cout, int Z, int Y, int X, float* out, void* workspace,
                             size_t workspace_bytes, int elem_type, int chunks, fbbev_stream_t stream);

/* The same step as ONE kernel (history_fused_x3_kernels.h): every MFMA wave blends the 8 trilinear taps of its own operands --
 * fbbev_history_warp_vm's taps, weights, order and rounding -- stores them to next[:, 1:] and feeds them to the split-operand
 * convolutions of fbbev_history_conv_bf16x3; the T warped frames are not read back.  next[:, 0] must hold the current frame.  The
 * ring and `out` are the same bits as the two calls.  C = Cout = 80, FBBEV_ELEM_BF16 / FBBEV_ELEM_F16, Z, Y, X >= 2; other shapes
 * FBBEV_E_UNSUPPORTED (use the two calls).  workspace: as fbbev_history_conv_bf16x3 with T1 = T + 1, plus 64 bytes. */
int fbbev_history_fused_x3_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                              const float* rt_flow, const float* w1, const float* bias1, const float* w2, const float* bias2,
                              int B, int T, int C, int Cout, int Z, int Y, int X, float* out, void* workspace,
                              size_t workspace_bytes, int elem_type, fbbev_stream_t stream);

/* ---- voxel-major history ring (opt-in layout of the inference ring; the reference's is (B, T*C, Z, Y, X), fbocc.py:234)
 * A frame is [voxel n = (z*Y + y)*X + x][channel]: the C elements of a voxel are contiguous, so a trilinear tap is a
 * 16-byte load of 8 (16-bit) / 4 (fp32) channels instead of one 2- / 4-byte gather per channel plane, and
 * fbbev_history_conv_bf16(voxel_major = 1) reads its operands as rows.  Element values are bit-identical to the planar
 * kernels'; only the addresses differ.  C % 8 == 0 (16-bit) or C % 4 == 0 (fp32); strides (elements) likewise; pointers
 * 16-byte aligned.
 *   fbbev_history_warp_vm : history (B, T, N, C) -> out (B, T, N, C), F.grid_sample(history, generate_grid(rt_flow)) of
 *                           fbocc.py:264-275 per frame; strides are per sample, 0 = dense.
 *   fbbev_history_frame_vm: curr (B, C, N) fp32 planes -> out[b] (N, C): slot 0 of the ring (fbocc.py:286 cat).
 *                           inner = 1: plane position = row; inner = Z: the planes are (Y, X, Z) volumes as the view
 *                           transformation returns them (fbocc.py:212 permutes to (Z, Y, X) first), rows are z-major. */
/* fbbev_history_conv_e (fp32 MFMA, the reference's arithmetic) on a voxel-major ring: feats (B, T1, N, C).  The K order
 * inside the fp32 accumulation differs from the planar kernel (equal to fp32 rounding).  C = Cout in {16, 80}; workspace
 * as fbbev_history_conv_e. */
int fbbev_history_conv_vm(const void* feats, long long feats_stride_b, const float* w1, const float* bias1, const float* w2,
                          const float* bias2, int B, int T1, int C, int Cout, int N, float* out, void* workspace,
                          size_t workspace_bytes, int elem_type, fbbev_stream_t stream);

/* Round 3 -- warp, new ring and both convolutions in ONE launch (the "fused warp-and-1x1x1-conv kernel" of SURVEY 8f-1;
 * fbocc.py:264-310 after the folding described above).  history (B,T,N,C) and `next` (B,T+1,N,C) are voxel-major 16-bit rings
 * (elem_type 1 = bf16, 2 = f16; C = Cout = 80; strides in elements, 0 = dense).  The caller stores the current frame into
 * slot 0 of `next` first (fbbev_history_frame_vm); this call writes slots 1..T (frame t+1 = history frame t re-sampled with
 * rt_flow: the SAME element bits as fbbev_history_warp_vm) and out (B,Cout,N) f32 = the fused volume on the bf16 MFMA with
 * fp32 accumulation -- the same operands and accumulation order as fbbev_history_conv_bf16 on that ring, without reading the
 * T new frames back.  workspace: as fbbev_history_conv_bf16.  Other shapes / an fp32 ring: FBBEV_E_UNSUPPORTED (use
 * fbbev_history_warp_vm + fbbev_history_conv_bf16). */
int fbbev_history_fused_vm(const void* history, long long history_stride_b, void* next, long long next_stride_b,
                           const float* rt_flow, const float* w1, const float* bias1, const float* w2,
                           const float* bias2, int B, int T, int C, int Cout, int Z, int Y, int X, float* out,
                           void* workspace, size_t workspace_bytes, int elem_type, fbbev_stream_t stream);
int fbbev_history_warp_vm(const void* history, long long history_stride_b, const float* rt_flow, int B, int T, int C, int Z,
                          int Y, int X, void* out, long long out_stride_b, int elem_type, fbbev_stream_t stream);
int fbbev_history_frame_vm(const float* curr, int B, int C, int N, int inner, void* out, long long out_stride_b,
                           int elem_type, fbbev_stream_t stream);

/* Dense 3-D convolution on NDHWC (torch channels_last_3d) f32 activations as an fp32-MFMA implicit GEMM, inference:
 * replaces the eval-mode Conv3d (+ folded BatchNorm) (+ residual) (+ ReLU) groups of CustomResNet3D (resnet3d.py:19-43,
 * 78-102), FPN3D (fpn3d.py:50-70) and OccHead (occupancy_head.py:82-141), which the reference runs in fp32 through the
 * vendor library.
 *   x (B,Di,Hi,Wi,Cin), out (B,Do,Ho,Wo,Cout) [transposed: (B,2Di,2Hi,2Wi,Cout)], residual like out or NULL.
 *   ksize 1, 2 or 3, one stride (1|2) and padding (0|1) for the three axes, Do = (Di + 2 pad - ksize) / stride + 1 (checked).
 *   transposed != 0: ConvTranspose3d kernel 2 stride 2 padding 0 (the head's deblock); ksize/stride/pad are ignored,
 *   Do,Ho,Wo must equal Di,Hi,Wi.
 *   weight_fragments: the weights in MFMA A-fragment order with the batch norm folded in,
 *     wf[parity][tap][j][mt][lane][e] = W[cout = 16mt + lane%16][cin = 16j + 4(lane/16) + e][tap],  zero for cout >= Cout,
 *     tap = (kd*k + kh)*k + kw, j < Cin/16, mt < ceil(Cout/16); parity = (a*2+b)*2+c of the output voxel for the transposed
 *     case (8 blocks, W[cin][cout][a][b][c]), one block otherwise.
 *   bias: ceil(Cout/16)*16 floats (zero padded).  Cin % 16 == 0, 16-byte aligned pointers, else FBBEV_E_UNSUPPORTED.
 * Exact fp32 arithmetic (v_mfma_f32_16x16x4_f32). */
int fbbev_conv3d_ndhwc(const float* x, const float* weight_fragments, const float* bias, const float* residual, int B,
                       int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksize, int stride, int pad,
                       int relu, int transposed, float* out, fbbev_stream_t stream);

/* 2-D convolution on NHWC (torch channels_last) f32 activations, inference: the same kernel on a one-plane volume with a
 * single tap along the plane axis; for the eval-mode Conv2d + folded BatchNorm (+ residual) (+ ReLU) groups of the image
 * backbone (mmdet ResNet bottlenecks, cfg fbocc-r50-cbgs_depth_16f_16x4_20e.py:119-129) and CustomFPN (necks/fpn.py:108-134).
 * weight_fragments: the 3-D layout of the weight viewed as (Cout, Cin, 1, k, k).  ksize 1 or 3, Cin % 16 == 0. */
int fbbev_conv2d_nhwc(const float* x, const float* weight_fragments, const float* bias, const float* residual, int B,
                      int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad, int relu,
                      float* out, fbbev_stream_t stream);

/* bf16-MFMA variant of fbbev_conv3d_ndhwc / fbbev_conv2d_nhwc (inference): same fp32 NDHWC activations, bias, residual and
 * output; inputs and weights are rounded to bf16 (nearest even) in front of v_mfma_f32_16x16x32_bf16, accumulation is
 * fp32 -- the execution option the detector's `*_dtype='bf16'` knobs select on the vendor route, here on the hand-written one.
 *   weight_fragments_bf16[parity][tap][j][mt][lane][e] = bf16(W[cout = 16mt + lane%16][cin = 32j + 8(lane/16) + e][tap]);
 *   planar != 0: x is one plane (Di = Do = 1) and the kernel has a single tap along that axis (the 2-D case).
 * ksize 1 or 3 (transposed: kernel 2 stride 2), Cin % 32 == 0, else FBBEV_E_UNSUPPORTED. */
int fbbev_conv3d_ndhwc_bf16(const float* x, const void* weight_fragments_bf16, const float* bias, const float* residual,
                            int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int ksize, int stride,
                            int pad, int relu, int transposed, int planar, float* out, fbbev_stream_t stream);

/* The 3x3x3, stride 1, padding 1 case of fbbev_conv3d_ndhwc_bf16 with the input halo staged in LDS: a workgroup owns a
 * 4x8x8 voxel tile and reads its 600 halo rows once per 32-channel group (2.3 reads of the input per 64-channel output
 * block instead of 27).  Same arguments / weight layout / results (up to fp32 summation order) as that entry point.
 * x, out, residual (B,D,H,W,C) f32; Cin % 32 == 0. */
int fbbev_conv3d_k3s1_tiled_bf16(const float* x, const void* weight_fragments_bf16, const float* bias, const float* residual,
                                 int B, int D, int H, int W, int Cin, int Cout, int relu, float* out, fbbev_stream_t stream);

/* Data gradient of fbbev_conv3d_ndhwc's convolution (training): dx[i] = sum_k W_k^T dy[(i + pad - k) / stride] over the
 * taps for which the division is exact.  dy (B,Do,Ho,Wo,Cout), dx (B,Di,Hi,Wi,Cin) with the forward geometry (checked);
 * weight_fragments_t = the fragment layout of the TRANSPOSED weight (Cin, Cout, k, k, k) -- same tap index;
 * zero_bias: ceil(Cin/16)*16 zeros.  Cout % 16 == 0.  (The data gradient of the head's ConvTranspose3d(k=2,s=2) is the
 * forward entry point itself with ksize 2, stride 2, pad 0 on the deconvolution weight read as (out=Cin, in=Cout,2,2,2).) */
int fbbev_conv3d_dgrad_ndhwc(const float* dy, const float* weight_fragments_t, const float* zero_bias, int B, int Do,
                             int Ho, int Wo, int Cout, int Di, int Hi, int Wi, int Cin, int ksize, int stride, int pad,
                             float* dx, fbbev_stream_t stream);

/* Weight gradient of fbbev_conv3d_ndhwc's convolution (training):
 *   dw[tap][cout][cin] += sum over output voxels v of dy[v][cout] * x[v * stride + tap - pad][cin]
 * x (B,Di,Hi,Wi,Cin), dy (B,Do,Ho,Wo,Cout) NDHWC f32 with the forward geometry (checked); dw (ksize^3, Cout, Cin) f32,
 * ZERO on entry (voxel chunks meet through fp32 atomic adds; tap = (kd*k + kh)*k + kw).  Cin % 4 == Cout % 4 == 0.
 * (For the head's ConvTranspose3d(k=2,s=2) call it with x := the fine output gradient, dy := the coarse input,
 *  ksize 2, stride 2, pad 0: dw[tap][cin_deconv][cout_deconv].) */
int fbbev_conv3d_wgrad_ndhwc(const float* x, const float* dy, int B, int Di, int Hi, int Wi, int Cin, int Do, int Ho,
                             int Wo, int Cout, int ksize, int stride, int pad, float* dw, fbbev_stream_t stream);

/* Soft-weighted multi-level blend of the occupancy head on NDHWC f32 (inference): replaces the F.interpolate(trilinear,
 * align_corners=False) + `out += feats * weights` loop of OccHead.forward_coarse_voxel (occupancy_head.py:159-170).
 *   out[b,v,:] = wsoft[b,v,0] * level0[b,v,:] + sum_k wsoft[b,v,k] * trilinear(coarse_k)[b,v,:],  k = 1..n_coarse (<= 3)
 *   level0, out (B,D,H,W,C); coarse_k (B, dims[3k], dims[3k+1], dims[3k+2], C); wsoft (B,D,H,W,K), K >= n_coarse + 1.
 *   `coarse` (n_coarse device pointers) and `coarse_dims` (3 * n_coarse ints) are HOST arrays, read before the launch.
 * C % 4 == 0 and 16-byte aligned pointers, else FBBEV_E_UNSUPPORTED. */
int fbbev_blend_levels_ndhwc(const float* level0, const float* const* coarse, const int* coarse_dims, int n_coarse,
                             const float* wsoft, int K, int B, int D, int H, int W, int C, float* out,
                             fbbev_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FBBEV_H_ */
