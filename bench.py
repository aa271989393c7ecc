#!/usr/bin/env python3
"""bench.py -- samples/s of the FB-OCC forward view-transformation hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (one rank per GPU, RCCL via backend "nccl"); rank 0 prints ONE JSON line.  Started WITHOUT a launcher
  (`python bench.py --gpus N`, WORLD_SIZE unset) it re-executes itself under torch.distributed.run with N ranks.

--mode forward (default, the headline metric) | train (BASELINE configs[3]: forward_train + backward of the whole
  FB-OCC detector, 4 samples per GPU, gradients of ALL parameters averaged over ranks by bucketed RCCL all-reduces
  launched from autograd hooks, clip + AdamW step; the one collective of the path).

Step   = one pass of the hot path (scope S2 of SURVEY 8d: get_lidar_coor + voxel ranking (fused) ->
         bev_pool_v2 into the dense (B,C,Z,Y,X) volume) over one batch of synthetic 6-camera samples,
         BASELINE.json configs[1] shapes (6x256x704 in, 16x44 feature map, D=59, C=80, 200x200x16 grid).
         Nothing is cached across steps: the index tensors are rebuilt every step, as the reference
         does (view_transformer.py:628 disables its acceleration path).  Inputs are resident in HBM.
Shard  = independent samples: each rank owns `--batch` samples (weak scaling), no data-path collective.
value  = total samples all ranks processed / max-over-ranks wall time of the K timed steps.
roofline = algorithmic HBM bytes of ONE bev_pool_v2 dense-forward launch (BASELINE.md section 4 /
         SURVEY 8d: depth + feat + 4*(3P+2I) + dense output, each once) / its mean launch duration
         from HIP events recorded on the launch stream inside the timed region, vs 8 TB/s.
cpu_baseline = the same scope in pure CPU PyTorch (oracle restatement of the reference ops), rank 0,
         N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


class no_gc:
    """Timed regions run with the cyclic garbage collector paused (collected first, re-enabled after): a full collection of a
    process that has imported torch takes 30-80 ms, and one of them landing inside a 30-step loop of 0.45 ms steps tripled a
    leg's ms_per_step (tools/diag_bf16_leg.py + tools/diag_bf16_leg_trace.sh: the GPU time line shows the kernels at their usual
    durations and ONE idle gap of 31-80 ms between two steps).  Nothing is skipped: reference-counted frees are unaffected."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *a):
        import gc
        if self.was:
            gc.enable()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU per step (weak scaling)')
    ap.add_argument('--config', default='BL2')
    ap.add_argument('--tile-voxels', type=int, default=None, help='default: chosen by grid density')
    ap.add_argument('--pool-flags', type=lambda x: int(x, 0), default=None)
    ap.add_argument('--storage', choices=['f32', 'bf16', 'f16'], default='f32',
                    help='element type the BEV volume is STORED in (sums are always fp32); the reference is f32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-storage', action='store_true', help='skip the extra bf16-storage leg of the forward mode')
    ap.add_argument('--pipeline', choices=['alternate', 'graphs'], default='graphs',
                    help='--streams > 1: launch the steps eagerly on alternating streams, or replay one captured hipGraph per stream')
    ap.add_argument('--streams', type=int, default=2,
                    help='forward mode, N=1, experiment: after the timed loop, time the same steps once more with consecutive '
                         'batches on this many HIP streams (the rank build of batch i+1 may overlap the pooling of batch i); '
                         'reported as the extra "pipelined" object, never as `value`.  Round 6: on by default (2 streams, hipGraph replay '
                         'per stream) so that the overlap of the latency-bound ranking chain with the pooling kernel is driver-timed '
                         '(VERDICT r5 item 6); 1 switches the leg off')
    ap.add_argument('--launch', choices=['auto', 'graph', 'eager'], default='auto',
                    help='forward mode: how the ~10 short launches of the index build (geometry, ranking, NCHW->NHWC, tile index) '
                         'reach the GPU in every step: replayed from ONE captured hipGraph (the step then needs 2 host launches '
                         'and stays GPU-bound on a slow or busy host CPU; ~1 %% slower than eager launches on a fast one) or launched '
                         'one by one; auto (default) = whichever ran the warm-up steps faster.  The pooling kernel is always '
                         'launched eagerly between the HIP events that time it')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-fb-projection', action='store_true',
                    help='forward mode, N=1: skip the extra `fb_projection` leg (BASELINE configs[2]: forward + backward projection)')
    ap.add_argument('--fb-steps', type=int, default=50, help='timed steps of the fb_projection leg')
    ap.add_argument('--no-reference-gpu', action='store_true',
                    help="forward mode, N=1: skip the `reference_same_gpu` leg (the reference's own kernel from oracle/_ref timed on this GPU)")
    ap.add_argument('--no-fb-train', action='store_true',
                    help='forward mode, N=1: skip the extra `fb_projection_train` leg (forward + backward of the path at configs[2] shapes)')
    ap.add_argument('--mode', choices=['forward', 'train'], default='forward')
    ap.add_argument('--sync-bn', action='store_true', help="train mode: cross-rank statistics for the config's SyncBN layers "
                    '(SURVEY 8e: off for the headline, the delta is reported separately)')
    ap.add_argument('--bucket-mb', type=int, default=64, help='train mode: gradient bucket size')
    ap.add_argument('--conv', choices=['vendor', 'mfma'], default='mfma',
                    help='train mode: 3-D convolution stacks on the vendor library or on fbbev_conv3d_* (fwd + dgrad + wgrad)')
    ap.add_argument('--no-alt-dtype', action='store_true', help='train mode: skip the second timing with the other --conv-dtype')
    ap.add_argument('--conv-dtype', choices=['f32', 'bf16'], default='bf16',
                    help='train mode: compute dtype of the 2-D stacks (image backbone / neck / depth net), channels-last on the vendor '
                         'library; default bf16 with fp32 master weights, gradients and every other block fp32 (stated in `dtype`)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`
    (one rank per GPU, rank -> GPU binding through LOCAL_RANK, rendezvous on 127.0.0.1)."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    import socket
    import torch
    have = torch.cuda.device_count()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = launch_command(args, have, port)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def launch_command(args, have, port, argv=None):
    """argv of the self-launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the original flags>` -- the command line the driver itself uses for N > 1."""
    if have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node')
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)


def check_ranks(args, world, rank, dev, index=None):
    """The job really is `--gpus` RCCL ranks, one per device: the process group's size (not just the environment) must equal
    --gpus, and every rank reports the device it is bound to (all-gathered, so rank 0 can print the binding and two ranks
    on one GPU are refused).  -> (number of ranks in the group, [device index per rank])."""
    import torch
    import torch.distributed as dist
    n = dist.get_world_size() if dist.is_initialized() else 1
    if n != args.gpus or n != world:
        raise SystemExit(f'bench.py --gpus {args.gpus}: the process group has {n} rank(s) (WORLD_SIZE={world})')
    mine = torch.tensor([index if index is not None else (dev.index or 0)], dtype=torch.int64, device=dev)
    if n == 1:
        return 1, [int(mine)]
    got = [torch.zeros_like(mine) for _ in range(n)]
    dist.all_gather(got, mine)
    devices = [int(t) for t in got]
    if len(set(devices)) != n:
        raise SystemExit(f'bench.py --gpus {args.gpus}: ranks share a device ({devices}); one rank per GPU is required')
    return n, devices


def cpu_baseline(cfg, seconds):
    """Pure-CPU PyTorch restatement of the same scope (oracle.ViewTransformerOracle + index_add
    pooling), timed on this host's cores on a bounded sample (1 sample per iteration)."""
    import torch
    from fb_bev_amd import synthetic as S
    from oracle import oracle as O
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, 1, seed=0, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, 1, seed=0)
    shape = ovt.bev_feat_shape(1, cfg.channels)

    def one():
        coor = ovt.get_lidar_coor(*cam)
        rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
        feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
        return O.bev_pool_v2_torch(depth, feat, rd, rf, rb, shape)
    # pick the thread count that is FASTEST for this small-op workload (all 256 hardware threads of the
    # GPU box's host are ~20x slower than 16-32 because of intra-op fork/join overhead)
    best = (None, 1e9)
    ncpu = os.cpu_count() or 1
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        one()
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (th, dt)
        if dt > 8.0:
            break
    torch.set_num_threads(best[0])
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    return {'value': n / dt, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} x 1-sample {cfg.name} passes in {dt:.1f}s, best of 8/16/32/64/all threads on {ncpu} hw threads (torch {torch.__version__} CPU ops: '
                      'inverse/matmul geometry, argsort ranking, index_add pooling, permute)'}


def main():
    args = parse()
    self_launch(args)                       # no-op under a launcher / for --gpus 1
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.mode == 'train':
        return run_train(args)
    return run_forward(args)


def cpu_baseline_c(cfg, seconds):
    """Second CPU baseline of SURVEY 8d: same scope, the pooling done by the loop-exact C oracle (OpenMP over
    intervals, oracle/fbbev_oracle.c) instead of torch index_add; geometry + ranking are the same torch CPU ops."""
    import torch
    from fb_bev_amd import synthetic as S
    from oracle import oracle as O
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, 1, seed=0, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, 1, seed=0)
    shape = ovt.bev_feat_shape(1, cfg.channels)

    def one():
        rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(ovt.get_lidar_coor(*cam))
        return O.bev_pool_v2(depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb, shape, st, ln)
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    return {'value': n / dt, 'unit': 'samples/s', 'cores': int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)),
            'kind': 'port', 'sample': f'{n} x 1-sample {cfg.name} passes in {dt:.1f}s; pooling = loop-exact C oracle with OpenMP over '
                                      f'intervals, geometry + ranking = torch CPU ops on {torch.get_num_threads()} threads'}


def cpu_baseline_fb(pc, levels, state, gcb, dbound, mlvl_shapes, seconds):
    """CPU baseline of the forward + backward projection scope (S3 of SURVEY 8d, BASELINE configs[2]): the oracle restatement
    on this host's cores, one sample per pass -- forward projection (torch CPU ops: geometry, argsort ranking, index_add
    pooling), Z-mean, oracle/backward_projection_oracle.py (BackwardProjection.forward restated: self-attention, depth-aware
    deformable cross-attention with its per-camera rebatch loops, FFN, LayerNorms) and the re-add."""
    import torch
    from fb_bev_amd import synthetic as S
    from oracle import backward_projection_oracle as BO, oracle as O
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(32, ncpu))
    ovt = O.ViewTransformerOracle(pc.grid_config, pc.input_size, pc.downsample)
    cam = S.camera_rig(pc, 1, seed=0, bda_aug=True)
    depth, ctx = S.depth_and_context(pc, 1, seed=0)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(1, pc.n_cams, pc.channels, h, w, generator=g) for h, w in mlvl_shapes]
    feats[0] = ctx
    X, Y, Z = pc.grid_xyz
    shape = ovt.bev_feat_shape(1, pc.channels)

    def one():
        rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(ovt.get_lidar_coor(*cam))
        vol = O.bev_pool_v2_torch(depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb, shape).permute(0, 1, 3, 4, 2)
        refined = BO.backward_projection(state, feats, vol.mean(-1), cam, depth, Y, X, gcb, pc.input_size, dbound,
                                         inverse=O.inv3x3_closed_form)
        return refined[..., None] + vol
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 50:
            break
    return {'value': n / dt, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} x 1-sample passes of the same scope in {dt:.1f}s on {torch.get_num_threads()} of {ncpu} hw threads: oracle forward '
                      f'projection (torch CPU ops) + Z-mean + oracle/backward_projection_oracle.py ({levels} levels, {Y}x{X} queries) + re-add'}


def fb_projection_leg(dev, steps, warmup, cpu_seconds, with_cpu=True):
    """BASELINE configs[2] on one GPU, reported beside `value` (never as it): forward projection + backward projection (BEV
    self-attention, depth-aware multi-scale deformable cross-attention over 6 cameras x 4 levels, FFN) + re-add for B = 4 samples
    of 200 x 200 BEV queries -- scope S3 of SURVEY 8d, the reference's tools/analysis_tools/benchmark_view_transformer.py:64-134
    protocol (indices rebuilt every step, nothing cached across steps except per-shape constants).  `ms_per_step` is the wall
    time of K back-to-back steps between device synchronisations; p10/p50/p90 and the DA kernel time are HIP events on the
    launch stream."""
    import torch
    from fb_bev_amd import _capi, configs, synthetic as S
    from fb_bev_amd.fb_view_transform import FBViewTransform
    from fb_bev_amd.graphed import Graphed

    def build(name, B, levels):
        pc = S.CONFIGS[name]
        X, Y, Z = pc.grid_xyz
        gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
        cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                                grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample,
                                num_levels=levels)
        torch.manual_seed(0)
        m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection'])
        with torch.no_grad():      # the reference init zeroes these heads: offsets / weights would not depend on the queries
            for n_, p_ in m.named_parameters():
                if 'sampling_offsets.weight' in n_ or 'attention_weights.weight' in n_:
                    p_.normal_(0, 0.05)
        m = m.to(dev).eval()
        cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
        depth, ctx = (t.to(dev) for t in S.depth_and_context(pc, B, seed=0))
        H, W = ctx.shape[-2:]
        shapes = [(H, W), (2 * H, 2 * W), (H // 2, W // 2), (H // 4, W // 4)][:levels]     # level 0 = the depth net's level
        g = torch.Generator().manual_seed(5)
        mlvl = [torch.randn(B, pc.n_cams, pc.channels, h, w, generator=g).to(dev) for h, w in shapes]
        mlvl[0] = ctx
        return pc, cfg, gcb, m, cam, depth, ctx, mlvl, shapes

    B, levels = 4, 4
    pc, cfg, gcb, m, cam, depth, ctx, mlvl, shapes = build('BL2', B, levels)
    X, Y, Z = pc.grid_xyz
    Q, E, ncam = X * Y, pc.channels, pc.n_cams
    da_ev = []
    real_da = _capi.da_cross_attn_fused

    def timed_da(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_da(*a, **k)
        e1.record()
        da_ev.append((e0, e1))
        return r

    out = {}
    with torch.no_grad():
        for _ in range(max(3, warmup)):
            m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize(dev)
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        _capi.da_cross_attn_fused = timed_da
        try:
            with no_gc():
                t0 = time.perf_counter()
                for i in range(steps):
                    sev[i][0].record()
                    res = m(cam, ctx, depth, mlvl_feats=mlvl)
                    sev[i][1].record()
                torch.cuda.synchronize(dev)
                elapsed = time.perf_counter() - t0
        finally:
            _capi.da_cross_attn_fused = real_da
        step_ms = sorted(a.elapsed_time(b) for a, b in sev)
        pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]  # noqa: E731
        da_ms = sum(a.elapsed_time(b) for a, b in da_ev) / len(da_ev) if da_ev else None
        # launches of one step (for the launch-count x 5 us floor SURVEY 8d prescribes for the attention part)
        launches = None
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                m(cam, ctx, depth, mlvl_feats=mlvl)
                torch.cuda.synchronize(dev)
            launches = sum(e.count for e in prof.key_averages() if e.device_time_total > 0)
        except Exception:
            launches = None
        # the same call replayed from a captured hipGraph
        graph_ms = None
        try:
            gr = Graphed(m, cam, ctx, depth, mlvl_feats=mlvl)
            gr(cam, ctx, depth, mlvl_feats=mlvl)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                gr(cam, ctx, depth, mlvl_feats=mlvl)
            torch.cuda.synchronize(dev)
            graph_ms = 1e3 * (time.perf_counter() - t1) / steps
            del gr
        except Exception as e:
            graph_ms = f'{type(e).__name__}: {e}'[:160]
        # A/B: the same step with every row-wise layer on the vendor fp32 GEMM (FBBEV_ROWS_LINEAR=f32; this also
        # takes the round-3 attention kernels, whose projections are separate GEMMs)
        f32_ms = None
        try:
            from fb_bev_amd import rows_linear as _RLmod
            was = _RLmod.X3
            _RLmod.X3 = False
            try:
                for _ in range(3):
                    m(cam, ctx, depth, mlvl_feats=mlvl)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(steps):
                    m(cam, ctx, depth, mlvl_feats=mlvl)
                torch.cuda.synchronize(dev)
                f32_ms = 1e3 * (time.perf_counter() - t1) / steps
            finally:
                _RLmod.X3 = was
        except Exception as e:
            f32_ms = f'{type(e).__name__}: {e}'[:160]
        S_tok = sum(h * w for h, w in shapes)
        Za = 4
        DC, Hd, Wd = depth.shape[2], depth.shape[3], depth.shape[4]
        parts = {'slots_written': 4 * B * Q * E, 'query_rows_read': 4 * B * Q * E + 4 * Q * E,
                 'reference_points_mask_depth_records': ncam * B * Q * Za * (8 + 1 + 4),
                 'camera_token_head_planes': 4 * B * ncam * S_tok * E, 'depth_distribution': 4 * B * ncam * DC * Hd * Wd}
        algo = sum(parts.values())
        ms = 1e3 * elapsed / steps
        out = {
            'what': 'BASELINE configs[2] (scope S3 of SURVEY 8d): forward projection + backward projection + re-add, 1 x MI355X',
            'value': B * steps / elapsed, 'unit': 'samples/s', 'ms_per_step': ms, 'steps': steps,
            'step_gpu_ms_p10_p50_p90': [pct(0.1), pct(0.5), pct(0.9)],
            'dtype': 'f32 storage/accumulate, bf16x3 split products in the row-wise layers (fp32-grade, ~1e-5 relative; FBBEV_ROWS_LINEAR=f32: vendor fp32 GEMMs)',
            'data': 'synthetic', 'hipgraph_replay_ms_per_step': graph_ms, 'fp32_gemm_route_ms': f32_ms,
            'config': {'workload': f'FB-OCC forward + backward projection, BASELINE configs[2]: 6x{pc.input_size[0]}x{pc.input_size[1]} in, D={DC}, C={E}, '
                                   f'grid {X}x{Y}x{Z}, {Y}x{X} BEV queries, {levels} attention levels '
                                   f'{"/".join(f"{h}x{w}" for h, w in shapes)}, 8 points, 4 Z anchors, 1 encoder layer; indices rebuilt every step',
                       'samples_per_gpu': B},
            'roofline': {'kernel': 'k_da_cross_attn_fused', 'bound': 'hbm', 'achieved': (algo / (da_ms * 1e-3) / 1e9) if da_ms else None,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': (algo / (da_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if da_ms else None,
                         'traffic': None, 'algorithmic_bytes_per_launch': algo, 'algorithmic_bytes_parts': parts, 'kernel_ms': da_ms,
                         'note': 'the sampler is bound by vector-L1 accesses / latency, not by HBM (SURVEY 8d: working set < L2); the HBM '
                                 'fraction is reported because the contract asks for it, the launch floor below is the second yardstick'},
            'launch_floor': {'launches_per_step': launches, 'us_per_launch': 5.0,
                             'floor_ms': None if launches is None else launches * 5e-3,
                             'step_over_floor': None if not launches else ms / (launches * 5e-3)},
        }
        del m, mlvl, res
        # the shipped shape (100 x 100 queries, one 16x44 level, D = 80), B = 1, replayed from a captured hipGraph
        try:
            pc1, _, _, m1, cam1, depth1, ctx1, _, _ = build('REF', 1, 1)
            for _ in range(3):
                m1(cam1, ctx1, depth1)
            g1 = Graphed(m1, cam1, ctx1, depth1)
            g1(cam1, ctx1, depth1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                g1(cam1, ctx1, depth1)
            torch.cuda.synchronize(dev)
            gms = 1e3 * (time.perf_counter() - t1) / steps
            t1 = time.perf_counter()
            for _ in range(steps):
                m1(cam1, ctx1, depth1)
            torch.cuda.synchronize(dev)
            ems = 1e3 * (time.perf_counter() - t1) / steps
            out['shipped_shape'] = {'workload': 'shipped fbocc-r50 shape: 100x100 queries, one 16x44 level, D=80, grid 100x100x8, B=1',
                                    'hipgraph_replay_ms_per_step': gms, 'eager_ms_per_step': ems, 'samples_per_s_graph': 1e3 / gms}
            del g1, m1
        except Exception as e:
            out['shipped_shape'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    if with_cpu:
        pcb, cfgb, gcbb, mb, *_ = build('BL2', 1, levels)
        state = {k: v.detach().cpu() for k, v in mb.backward_projection.state_dict().items()}
        del mb
        out['cpu_baseline'] = cpu_baseline_fb(pcb, levels, state, gcbb, cfgb['depth_bound'], shapes, cpu_seconds)
        out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
    return out


def cpu_baseline_fb_train(pc, levels, state, gcb, dbound, mlvl_shapes, seconds):
    """CPU baseline of the path's TRAINING step (BASELINE configs[2] / [3] scope): the oracle restatement of cpu_baseline_fb under
    torch CPU autograd, one sample per pass -- forward as there, then backward from an upstream gradient in the output's shape to
    depth, context, the pyramid levels and every parameter of the backward projection."""
    import torch
    from fb_bev_amd import synthetic as S
    from oracle import backward_projection_oracle as BO, oracle as O
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(32, ncpu))
    ovt = O.ViewTransformerOracle(pc.grid_config, pc.input_size, pc.downsample)
    cam = S.camera_rig(pc, 1, seed=0, bda_aug=True)
    depth0, ctx0 = S.depth_and_context(pc, 1, seed=0)
    g = torch.Generator().manual_seed(5)
    feats0 = [torch.randn(1, pc.n_cams, pc.channels, h, w, generator=g) for h, w in mlvl_shapes]
    X, Y, Z = pc.grid_xyz
    shape = ovt.bev_feat_shape(1, pc.channels)
    gout = torch.randn(1, pc.channels, Y, X, Z, generator=torch.Generator().manual_seed(11))

    def one():
        P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in state.items()}
        depth, ctx = depth0.clone().requires_grad_(), ctx0.clone().requires_grad_()
        feats = [ctx] + [f.clone().requires_grad_() for f in feats0[1:]]
        with torch.no_grad():
            rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(ovt.get_lidar_coor(*cam))
        vol = O.bev_pool_v2_torch(depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb, shape).permute(0, 1, 3, 4, 2)
        refined = BO.backward_projection(P, feats, vol.mean(-1), cam, depth, Y, X, gcb, pc.input_size, dbound,
                                         inverse=O.inv3x3_closed_form)
        (refined[..., None] + vol).backward(gout)
        return depth.grad
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 50:
            break
    return {'value': n / dt, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} x 1-sample forward+backward passes of the same scope in {dt:.1f}s on {torch.get_num_threads()} of {ncpu} hw threads: '
                      f'the oracle (forward projection by torch CPU ops, oracle/backward_projection_oracle.py, {levels} levels, {Y}x{X} '
                      f'queries, re-add) under torch CPU autograd'}


def fb_projection_train_leg(dev, steps, warmup, cpu_seconds, with_cpu=True):
    """The path's TRAINING step on one GPU, reported beside `value` (never as it): forward + backward of FBViewTransform (forward
    projection, Z-mean, backward projection with both deformable attentions, FFN, LayerNorms, re-add) at BASELINE configs[2] shapes,
    B = 4 -- the per-GPU share of configs[3]'s batch 32 on 8 GPUs -- with the upstream gradient handed over in the output's own layout
    (what the voxel encoder's backward gives the path).  Gradients of depth, context, every pyramid level and every parameter are
    produced; nothing is cached across steps.  Reference: bev_pool.py:40-80, bev_pool_cuda.cu:64-118,
    multi_scale_deformable_attn_function.py:137-172, bevformer_encoder.py:206-377 under autograd.  `ms_per_step` = wall time of K
    back-to-back steps between device synchronisations; p10/p50/p90 are HIP events on the launch stream; the DA backward (hit lists +
    unit gradients + output-owned value-gradient planes) is bracketed by HIP events, its kernels split by a one-step roctracer profile."""
    import torch
    from fb_bev_amd import _capi, synthetic as S, train_path as TP
    B, levels = 4, 4
    d = S.fb_path_step('BL2', B, levels, dev)
    pc, step, shapes = d['pc'], d['step'], d['shapes']
    X, Y, Z = pc.grid_xyz
    Q, E, ncam = X * Y, pc.channels, pc.n_cams
    for _ in range(max(3, warmup)):
        step()
    torch.cuda.synchronize(dev)
    da_ev = []
    reals = {n: getattr(_capi, n) for n in ('da_cross_attn_bwd', 'da_cross_attn_bwd_planes')}     # whichever entry the route takes

    def timed_of(real):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real(*a, **k)
            e1.record()
            da_ev.append((e0, e1))
            return r
        return timed
    sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for n, f in reals.items():
        setattr(_capi, n, timed_of(f))
    try:
        with no_gc():
            t0 = time.perf_counter()
            for i in range(steps):
                sev[i][0].record()
                step()
                sev[i][1].record()
            torch.cuda.synchronize(dev)
            elapsed = time.perf_counter() - t0
    finally:
        for n, f in reals.items():
            setattr(_capi, n, f)
    step_ms = sorted(a.elapsed_time(b) for a, b in sev)
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]  # noqa: E731
    da_ms = sum(a.elapsed_time(b) for a, b in da_ev) / len(da_ev) if da_ev else None
    # forward only (autograd graph recorded), same inputs
    m, cam, ctx, depth, mlvl = d['model'], d['cam'], d['ctx'], d['depth'], d['mlvl']
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(steps):
        o = m(cam, ctx, depth, mlvl_feats=mlvl)
    torch.cuda.synchronize(dev)
    fwd_ms = 1e3 * (time.perf_counter() - t1) / steps
    del o
    kernels, launches = {}, None
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize(dev)
        ka = [e for e in prof.key_averages() if e.device_time_total > 0]
        launches = sum(e.count for e in ka)
        for e in ka:
            for key in ('k_da_bwd_scatter_owned', 'k_da_bwd_unit_planes', 'k_da_bwd_hitlist', 'k_msda_bwd_scatter', 'k_msda_bwd_unit',
                        'k_da_cross_attn_fused', 'k_rows_linear_x3', 'k_rows_wgrad_x3', 'k_pool_bwd_rows', 'Cijk_'):
                if key in e.key:
                    kernels[key] = kernels.get(key, 0.0) + e.device_time_total / 1e3
    except Exception:
        pass
    S_tok = sum(h * w for h, w in shapes)
    L, P, M, HS = levels, 8, 8, 12
    # algorithmic bytes of k_da_bwd_scatter_owned: hit records (64 B per (camera, query) hit slot), the upstream slot
    # gradient, the offsets / attention words of every unit, the value gradient written once
    parts = {'hit_records': ncam * B * Q * 64, 'grad_slots_read': 4 * B * Q * E, 'offsets_read': 4 * B * Q * M * L * P * 2,
             'attention_read': 4 * B * Q * M * L * P, 'grad_value_written': 4 * B * ncam * S_tok * M * HS}
    algo = sum(parts.values())
    sc_ms = kernels.get('k_da_bwd_scatter_owned')
    ms = 1e3 * elapsed / steps
    out = {
        'what': 'training step of the PATH (forward + backward of the forward-backward view transformation) at BASELINE configs[2] shapes, '
                'B = 4 = the per-GPU share of configs[3]; upstream gradient handed over in the output layout; 1 x MI355X',
        'value': B * steps / elapsed, 'unit': 'samples/s', 'ms_per_step': ms, 'steps': steps,
        'step_gpu_ms_p10_p50_p90': [pct(0.1), pct(0.5), pct(0.9)], 'forward_ms_train_mode': fwd_ms,
        'dtype': 'f32 storage/accumulate, bf16x3 split products in the row-wise layers (fp32-grade, ~1e-5 relative)', 'data': 'synthetic',
        'route': 'one autograd node per encoder layer on the inference kernels (FBBEV_TRAIN_FUSED)' if TP.TRAIN_FUSED else 'composite autograd (FBBEV_TRAIN_FUSED=0)',
        'config': {'workload': f'FB-OCC forward-backward view transformation, training step: 6x{pc.input_size[0]}x{pc.input_size[1]} in, '
                               f'D={depth.shape[2]}, C={E}, grid {X}x{Y}x{Z}, {Y}x{X} BEV queries, {levels} attention levels '
                               f'{"/".join(f"{h}x{w}" for h, w in shapes)}, 8 points, 4 Z anchors, 1 encoder layer; indices rebuilt every step; '
                               f'gradients for depth, context, all pyramid levels and all parameters',
                   'samples_per_gpu': B},
        'da_backward_ms_hip_events': da_ms, 'kernel_ms_one_step_profile': kernels, 'launches_per_step': launches,
        'roofline': {'kernel': 'k_da_bwd_scatter_owned', 'bound': 'hbm', 'achieved': (algo / (sc_ms * 1e-3) / 1e9) if sc_ms else None,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': (algo / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sc_ms else None,
                     'traffic': None, 'algorithmic_bytes_per_launch': algo, 'algorithmic_bytes_parts': parts, 'kernel_ms': sc_ms,
                     'note': 'the value-gradient scatter is bound by its 64-bit LDS atomics (one ds_add_u64 per channel and corner) and '
                             'the hit-list walk, not by HBM: counters in profiles/r06_*pmc_train*.json, DESIGN 3.3'},
    }
    del d, m, step
    torch.cuda.empty_cache()
    if with_cpu:
        db = S.fb_path_step('BL2', 1, levels, dev, train=False)
        state = {k: v.detach().cpu() for k, v in db['model'].backward_projection.state_dict().items()}
        cfgb, gcbb, pcb = db['cfg'], db['gcb'], db['pc']
        del db
        out['cpu_baseline'] = cpu_baseline_fb_train(pcb, levels, state, gcbb, cfgb['depth_bound'], shapes, cpu_seconds)
        out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
    return out


def reference_same_gpu_leg(dev, cfg, B, cam, depth, ctx, vt, idx, steps=10):
    """Stated baseline (never the target, never imported by the product): the REFERENCE's own bev_pool kernel --
    mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu compiled unmodified for gfx950 by oracle/Makefile into oracle/_ref -- timed on THIS
    GPU on the bench inputs, with the host-side ops the reference wraps around it, HIP events on the launch stream (the reference
    launches on the legacy default stream, which is torch's current stream here), outside every timed region:
      S1 = feat.permute().contiguous() (view_transformer.py:536 + bev_pool.py:18) + out = new_zeros (bev_pool.py:24) + bev_pool_v2
           kernel (bev_pool_cuda.cu:122-128) + x.permute(0, 4, 1, 2, 3).contiguous() (bev_pool.py:88), on exact-size index tensors;
      S2 = get_lidar_coor + voxel_pooling_prepare_v2 restated op for op with ATen on the GPU (view_transformer.py:458-498, 547-605:
           arange, sub / div, long, cat, the in-grid mask, boolean indexing, the fp32 rank, argsort, torch.where -- its device
           syncs included) + S1.
    Returns None when oracle/_ref is absent."""
    import ctypes
    import torch
    path = os.path.join(ROOT, 'oracle', '_ref', 'libbev_pool_ref.so')
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    fwd = getattr(lib, '_Z11bev_pool_v2iiPKfS0_PKiS2_S2_S2_S2_Pf')              # bev_pool_v2(int c, int n_intervals, ...), bev_pool_cuda.cu:122
    fwd.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
    fwd.restype = None
    Z, Y, X = vt.grid_zyx
    C = ctx.shape[2]
    rb, rd, rf, st, ln = [t.contiguous() for t in idx.exact()]
    lo, it, gs = (torch.tensor(v, dtype=torch.float32, device=dev) for v in vt._grid3())
    frustum = vt._frustum_dev(dev)
    vptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731

    def pool(rb_, rd_, rf_, st_, ln_):
        feat = ctx.permute(0, 1, 3, 4, 2).contiguous().float()
        out = feat.new_zeros((B, Z, Y, X, C))
        fwd(C, st_.numel(), vptr(depth), vptr(feat), vptr(rd_), vptr(rf_), vptr(rb_), vptr(st_), vptr(ln_), vptr(out))
        return out.permute(0, 4, 1, 2, 3).contiguous()

    def prepare():
        rots, trans, intr, post_rots, post_trans, bda = cam
        N = trans.shape[1]
        pts = frustum.to(rots) - post_trans.view(B, N, 1, 1, 1, 3)
        pts = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
        pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
        comb = rots.matmul(torch.inverse(intr))
        pts = comb.view(B, N, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1) + trans.view(B, N, 1, 1, 1, 3)
        coor = bda.view(B, 1, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1)).squeeze(-1)
        _, _, D, H, W, _ = coor.shape
        n = B * N * D * H * W
        r_d = torch.arange(0, n, dtype=torch.int, device=dev)
        r_f = torch.arange(0, n // D, dtype=torch.int, device=dev).reshape(B, N, 1, H, W).expand(B, N, D, H, W).flatten()
        vox = ((coor - lo) / it).long().view(n, 3)
        bidx = torch.arange(0, B).reshape(B, 1).expand(B, n // B).reshape(n, 1).to(vox)
        vox = torch.cat((vox, bidx), 1)
        kept = (vox[:, 0] >= 0) & (vox[:, 0] < gs[0]) & (vox[:, 1] >= 0) & (vox[:, 1] < gs[1]) & (vox[:, 2] >= 0) & (vox[:, 2] < gs[2])
        vox, r_d, r_f = vox[kept], r_d[kept], r_f[kept]
        r_b = vox[:, 3] * (gs[2] * gs[1] * gs[0])
        r_b += vox[:, 2] * (gs[1] * gs[0])
        r_b += vox[:, 1] * gs[0] + vox[:, 0]
        order = r_b.argsort()
        r_b, r_d, r_f = r_b[order], r_d[order], r_f[order]
        k2 = torch.ones(r_b.shape[0], device=dev, dtype=torch.bool)
        k2[1:] = r_b[1:] != r_b[:-1]
        s_ = torch.where(k2)[0].int()
        l_ = torch.zeros_like(s_)
        l_[:-1] = s_[1:] - s_[:-1]
        l_[-1] = r_b.shape[0] - s_[-1]
        return r_b.int().contiguous(), r_d.int().contiguous(), r_f.int().contiguous(), s_.int().contiguous(), l_.int().contiguous()

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            torch.cuda.synchronize(dev)
            ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2], r

    with torch.no_grad():
        s1_ms, vol1 = timed(lambda: pool(rb, rd, rf, st, ln))
        s2_ms, vol2 = timed(lambda: pool(*prepare()))
        # kernel alone (its share of S1), and a sanity check of both volumes against each other (argsort is unstable: sums may re-associate)
        feat = ctx.permute(0, 1, 3, 4, 2).contiguous().float()
        outk = feat.new_zeros((B, Z, Y, X, C))
        k_ms, _ = timed(lambda: fwd(C, st.numel(), vptr(depth), vptr(feat), vptr(rd), vptr(rf), vptr(rb), vptr(st), vptr(ln), vptr(outk)))
        close = bool(torch.allclose(vol1, vol2, rtol=1e-4, atol=1e-5))
    return {'what': 'the reference bev_pool_v2 kernel (bev_pool_cuda.cu compiled unmodified for gfx950, oracle/_ref) and the ATen ops the '
                    'reference wraps around it, on this GPU, same inputs; a stated baseline, not a target',
            's1_ms': s1_ms, 's2_ms': s2_ms, 'kernel_only_ms': k_ms, 'samples_per_s_s1': B / (s1_ms * 1e-3),
            'samples_per_s': B / (s2_ms * 1e-3), 'unit': 'samples/s (S2: index tensors rebuilt every step, as the reference does)',
            'timing': f'median of {steps} HIP-event intervals per scope, each followed by a device synchronisation',
            's1_equals_s2_volume_within_1e-4': close}


def device_identity(dev):
    """Which physical GPU this process drives, for the reader of a driver-side utilisation sample (VERDICT r5 hygiene: BENCH_r05's
    `gpu_busy` read 0 % on card0 while the kernel traces prove the work ran): marketing name, PCI bus id, the sysfs card that carries
    that bus id and its gpu_busy_percent read WHILE warm-up steps are queued (0 here too would mean the sampler, not the run)."""
    import glob
    import torch
    out = {}
    try:
        pr = torch.cuda.get_device_properties(dev)
        out['name'] = pr.name
        out['compute_units'] = getattr(pr, 'multi_processor_count', None)
        dom, bus, devid = getattr(pr, 'pci_domain_id', None), getattr(pr, 'pci_bus_id', None), getattr(pr, 'pci_device_id', None)
        if bus is not None:
            out['pci_bus_id'] = f'{dom or 0:04x}:{bus:02x}:{devid or 0:02x}.0'
        for card in sorted(glob.glob('/sys/class/drm/card[0-9]*')):
            try:
                addr = os.path.basename(os.path.realpath(os.path.join(card, 'device')))
            except OSError:
                continue
            if out.get('pci_bus_id') and addr.lower() == out['pci_bus_id'].lower():
                out['sysfs_card'] = os.path.basename(card)
                try:
                    out['gpu_busy_percent_during_warmup'] = int(open(os.path.join(card, 'device', 'gpu_busy_percent')).read().strip())
                except (OSError, ValueError):
                    pass
        out['visible_devices_env'] = {k: os.environ[k] for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES') if k in os.environ}
    except Exception as e:
        out['error'] = f'{type(e).__name__}: {e}'[:160]
    return out


def run_forward(args):
    import torch
    import torch.distributed as dist
    from fb_bev_amd import _capi
    from fb_bev_amd import synthetic as S
    from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D

    from fb_bev_amd import shard
    world, rank, local_rank = shard.world()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    shard.init('nccl', dev)     # backend "nccl" is RCCL on ROCm; used for the fence + max-reduce only
    ranks, devices = check_ranks(args, world, rank, dev)

    cfg = S.CONFIGS[args.config]
    B = args.batch
    # every rank gets its own samples (different seeds => different rigs/augmentations)
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=shard.shard_seed(rank), bda_aug=True)]
    depth, ctx = S.depth_and_context(cfg, B, seed=shard.shard_seed(rank))
    depth, ctx = depth.to(dev), ctx.to(dev)
    store_dt = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[args.storage]
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, tile_voxels=args.tile_voxels,
                                      pool_flags=args.pool_flags, out_dtype=store_dt).to(dev)
    args.tile_voxels, flags = vt.tiling(cfg.n_cams)            # density-aware tiling of this rig (6 cameras here)
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    tile_ws = vt._tile_ws(dev, B, args.tile_voxels)
    esz = 4 if args.storage == 'f32' else 2
    out = torch.empty((B, C, Z, Y, X), dtype=store_dt, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]       # whole-step GPU time, for the p10 / p50 / p90 spread

    def prep():
        idx = vt.build_index_from_cams(*cam)                        # fbbev_lift_rank_build: geometry + ranking, device counts
        feat = _capi.nchw_to_nhwc(ctx)                              # (B,N,H,W,C), the copy of bev_pool.py:18
        _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X,
                              tile_ws, args.tile_voxels)
        return idx, feat

    def pool(idx, feat, dst):
        _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,
                                    idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, dst, tile_ws,
                                    args.tile_voxels, flags)

    graph, graph_error = None, None
    if args.launch in ('graph', 'auto'):
        # The index build is ~10 launches of 5-45 us each: issued one by one they cost the host 0.2-1.2 ms per step depending
        # on the box's CPU -- more than the 0.7 ms the GPU needs on a slow or busy host.  Captured once, replayed per step:
        # the SAME kernels on the same buffers (indices still rebuilt from the camera tensors every step, device-side counts,
        # no host sync anywhere), one host launch.
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            prep()
        torch.cuda.synchronize(dev)
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of the process group may query events while this thread captures
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                g_idx, g_feat = prep()
            torch.cuda.synchronize(dev)
        except Exception as e:                                      # capture refused: the eager step is always available
            if args.launch == 'graph':
                raise
            graph, graph_error = None, f'{type(e).__name__}: {e}'[:200]
            torch.cuda.synchronize(dev)

    def step(i=None):
        if i is not None:
            sev[i][0].record()
        if graph is not None:
            graph.replay()
            idx, feat = g_idx, g_feat
        else:
            idx, feat = prep()
        if i is not None:
            ev[i][0].record()
        pool(idx, feat, out)
        if i is not None:
            ev[i][1].record()
            sev[i][1].record()
        return idx

    def fence():
        shard.fence(dev)

    launch_probe = None
    if args.launch == 'auto' and graph is not None:
        # warm-up doubles as the probe: W steps launched eagerly, W steps with the index build replayed from the graph, each
        # block wall-clocked between fences; the timed region uses the faster one (every rank decides from the max over ranks)
        captured, rates = graph, {}
        for mode in ('eager', 'graph'):
            graph = captured if mode == 'graph' else None
            step(); fence()
            tw = time.perf_counter()
            for _ in range(max(1, args.warmup)):
                step()
            fence()
            rates[mode] = shard.max_over_ranks(time.perf_counter() - tw, dev) / max(1, args.warmup)
        graph = captured if rates['graph'] <= rates['eager'] else None
        launch_probe = {'eager_ms_per_step': 1e3 * rates['eager'], 'graph_ms_per_step': 1e3 * rates['graph'],
                        'chosen': 'graph' if graph is not None else 'eager'}
    else:
        for _ in range(args.warmup):
            idx = step()
    fence()
    dev_id = None
    if rank == 0:                               # outside the timed region: queue a burst of steps and read the card's own busy counter under it
        for _ in range(20):
            step()
        dev_id = device_identity(dev)
        fence()
    with no_gc():
        t0 = time.perf_counter()
        for i in range(args.steps):
            idx = step(i)
        fence()
        elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dev)

    # the graph-replayed step must produce what the eagerly launched step produces: one eager step into a second volume
    graph_equal = None
    if graph is not None:
        ei, ef = prep()
        chk = torch.empty_like(out)
        pool(ei, ef, chk)
        fence()
        graph_equal = bool(torch.equal(chk, out))
        del chk, ei, ef

    # Extra leg (reported beside `value`, never as it): the same K steps with the volume STORED in bf16 -- the storage
    # dtype BASELINE configs[1] names; the per-voxel sums stay the fp32 in-order fmaf chains, rounded once at the store.
    # The default line stays fp32 because that is what the reference's op computes and stores (bev_pool.py:16-22) and what
    # the bit-exact parity bar is defined on.
    alt = None
    if world == 1 and args.storage == 'f32' and not args.no_alt_storage:
        v16 = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, out_dtype=torch.bfloat16).to(dev)
        tv16, fl16 = v16.tiling(cfg.n_cams)
        tws16 = v16._tile_ws(dev, B, tv16)
        out16 = torch.empty((B, C, Z, Y, X), dtype=torch.bfloat16, device=dev)
        ev16 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

        def step16(i=None):
            ix = v16.build_index_from_cams(*cam)
            ft = _capi.nchw_to_nhwc(ctx)
            _capi.pool_tile_index(ix.interval_rank, ix.interval_starts, ix.counts, ix.n, B, Z, Y, X, tws16, tv16)
            if i is not None:
                ev16[i][0].record()
            _capi.bev_pool_v2_dense_fwd(depth, ft, ix.ranks_depth, ix.ranks_feat, ix.interval_rank, ix.interval_starts,
                                        ix.interval_lengths, B, C, Z, Y, X, out16, tws16, tv16, fl16)
            if i is not None:
                ev16[i][1].record()
            return ix
        for _ in range(args.warmup):
            step16()
        fence()
        with no_gc():
            t16 = time.perf_counter()
            for i in range(args.steps):
                ix16 = step16(i)
            fence()
            t16 = time.perf_counter() - t16
        k16 = sum(a.elapsed_time(b) for a, b in ev16) / max(1, args.steps)
        P16, I16 = ix16.counts.tolist()
        ab16 = 4 * B * cfg.n_cams * cfg.D * cfg.feat_hw[0] * cfg.feat_hw[1] + 4 * B * cfg.n_cams * cfg.feat_hw[0] * cfg.feat_hw[1] * C + \
            4 * (3 * P16 + 2 * I16) + 2 * B * Z * Y * X * C
        same = torch.equal(out16, out.to(torch.bfloat16))           # == the fp32 volume rounded once
        # the bf16 instantiation's own floors: stores alone / all but the gathers / all but the stores
        fl16_ms = {}
        ft16 = _capi.nchw_to_nhwc(ctx)
        scratch16 = torch.empty_like(out16)
        names16 = {0: 'kernel_interleaved_ms', 1: 'store_floor_ms', 2: 'no_gather_ms', 3: 'no_store_ms'}
        ts16 = {m_: [] for m_ in names16}
        try:                                                            # interleaved K, 1, 2, 3, K, ...; medians (see the fp32 probe below)
            for it in range(17):
                for mode in (0, 1, 2, 3):
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record()
                    if mode == 0:
                        _capi.bev_pool_v2_dense_fwd(depth, ft16, ix16.ranks_depth, ix16.ranks_feat, ix16.interval_rank, ix16.interval_starts,
                                                    ix16.interval_lengths, B, C, Z, Y, X, scratch16, tws16, tv16, fl16)
                    else:
                        _capi.diag_pool_store_floor(depth, ft16, ix16.ranks_depth, ix16.ranks_feat,
                                                    ix16.interval_rank, ix16.interval_starts, ix16.interval_lengths, B, C, Z, Y, X,
                                                    scratch16, tws16, tv16, fl16, mode)
                    b_.record()
                    torch.cuda.synchronize(dev)
                    if it >= 2:
                        ts16[mode].append(a_.elapsed_time(b_))
            for m_, name in names16.items():
                fl16_ms[name] = sorted(ts16[m_])[len(ts16[m_]) // 2]
        except _capi.FbbevError:
            pass
        del scratch16, ft16
        alt = {'volume_storage': 'bf16', 'accumulate_dtype': 'f32', 'value': B * args.steps / t16, 'unit': 'samples/s',
               'ms_per_step': 1e3 * t16 / args.steps, 'tile_voxels': tv16, 'kernel_ms': k16,
               'algorithmic_bytes_per_launch': ab16, 'roofline_frac': ab16 / (k16 * 1e-3) / 1e9 / HBM_PEAK_GBS if k16 > 0 else None,
               'equals_fp32_volume_rounded_once': bool(same), 'store_floor_ms': fl16_ms.get('store_floor_ms'),
               'no_gather_ms': fl16_ms.get('no_gather_ms'), 'no_store_ms': fl16_ms.get('no_store_ms'),
               'kernel_interleaved_ms': fl16_ms.get('kernel_interleaved_ms'),
               'store_floor_over_kernel': (fl16_ms['store_floor_ms'] / fl16_ms['kernel_interleaved_ms'])
               if fl16_ms.get('store_floor_ms') and fl16_ms.get('kernel_interleaved_ms') else None}
        del out16, tws16, v16

    # Extra leg (reported beside `value`, not as it): consecutive batches on alternating HIP streams, each with its own index
    # workspace and output volume.  The ranking kernels are short latency-bound launches (<= 256 workgroups) that leave
    # most of the chip idle; on a second stream they run under the HBM-bound pooling kernel of the previous batch.
    piped = None
    if world == 1 and args.streams > 1:
        ns = args.streams
        ctxs = [(vt, tile_ws, out)]
        for _ in range(ns - 1):
            v2 = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, tile_voxels=args.tile_voxels,
                                              pool_flags=args.pool_flags, out_dtype=store_dt).to(dev)
            v2.tiling(cfg.n_cams)
            ctxs.append((v2, v2._tile_ws(dev, B, args.tile_voxels), torch.empty_like(out)))
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]

        def pstep(i):
            v, tws, o = ctxs[i % ns]
            with torch.cuda.stream(streams[i % ns]):
                ix = v.build_index_from_cams(*cam)
                ft = _capi.nchw_to_nhwc(ctx)
                _capi.pool_tile_index(ix.interval_rank, ix.interval_starts, ix.counts, ix.n, B, Z, Y, X, tws, args.tile_voxels)
                _capi.bev_pool_v2_dense_fwd(depth, ft, ix.ranks_depth, ix.ranks_feat, ix.interval_rank, ix.interval_starts,
                                            ix.interval_lengths, B, C, Z, Y, X, o, tws, args.tile_voxels, flags)

        if args.pipeline == 'graphs':
            # every stream's step captured once into a hipGraph (the whole step is capturable: no host sync, device-side
            # counts), replayed alternately: no per-kernel host launch cost left in the loop
            graphs = []
            for k in range(ns):
                fence()
                pstep(k)
                fence()
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, stream=streams[k]):
                    v, tws, o = ctxs[k]
                    ix = v.build_index_from_cams(*cam)
                    ft = _capi.nchw_to_nhwc(ctx)
                    _capi.pool_tile_index(ix.interval_rank, ix.interval_starts, ix.counts, ix.n, B, Z, Y, X, tws, args.tile_voxels)
                    _capi.bev_pool_v2_dense_fwd(depth, ft, ix.ranks_depth, ix.ranks_feat, ix.interval_rank, ix.interval_starts,
                                                ix.interval_lengths, B, C, Z, Y, X, o, tws, args.tile_voxels, flags)
                graphs.append(gk)

            def pstep(i):  # noqa: F811
                with torch.cuda.stream(streams[i % ns]):
                    graphs[i % ns].replay()

        fence()
        for i in range(max(args.warmup, 2 * ns)):
            pstep(i)
        fence()
        tp = time.perf_counter()
        for i in range(args.steps):
            pstep(i)
        fence()
        tp = time.perf_counter() - tp
        same = all(torch.equal(o, out) for _, _, o in ctxs[1:])     # every stream's volume == the single-stream one
        piped = {'streams': ns, 'scheme': args.pipeline, 'value': B * args.steps / tp, 'unit': 'samples/s', 'ms_per_step': 1e3 * tp / args.steps,
                 'volumes_identical_to_single_stream': bool(same),
                 'what': 'the same K steps, batch i on stream i % streams with its own index workspace and output volume: the '
                         'rank build of batch i+1 runs under the pooling kernel of batch i'}
        del ctxs, streams


    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    step_ms = sorted(a.elapsed_time(b) for a, b in sev)
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))] if step_ms else None  # noqa: E731
    # context for the roofline fraction (outside the timed region): what a plain device fill of the same output
    # buffer reaches on this box -- the practical write ceiling (report vs peak AND vs measured bandwidth)
    fill_ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out.zero_()
        b.record()
        torch.cuda.synchronize(dev)
        fill_ms.append(a.elapsed_time(b))
    fill_gbs = out.numel() * esz / (sorted(fill_ms)[len(fill_ms) // 2] * 1e-3) / 1e9
    # the kernel's own STORE PATTERN as a floor: the same instantiation, grid, tile walk, XCD order and
    # `sc1 nt` stores with the gathers compiled out (mode 1: stores alone; mode 2: all but the depth / feature gathers),
    # timed per launch with HIP events like the real kernel; outside the timed region, results never used
    floors = {}
    if args.storage == 'f32':
        feat_f = _capi.nchw_to_nhwc(ctx)
        scratch = torch.empty_like(out)
        # round 6: the real kernel and its three diagnostic instantiations are launched INTERLEAVED (K, 1, 2, 3, K, 1,
        # ...) and every one is reported by its MEDIAN, so that a drift of the box between two separate loops can no longer put the
        # stores-only floor above the kernel it bounds
        names = {0: 'kernel_interleaved_ms', 1: 'store_floor_ms', 2: 'no_gather_ms', 3: 'no_store_ms'}
        ts = {m_: [] for m_ in names}
        try:
            for it in range(17):
                for mode in (0, 1, 2, 3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if mode == 0:
                        _capi.bev_pool_v2_dense_fwd(depth, feat_f, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                                    idx.interval_lengths, B, C, Z, Y, X, scratch, tile_ws, args.tile_voxels, flags)
                    else:
                        _capi.diag_pool_store_floor(depth, feat_f, idx.ranks_depth, idx.ranks_feat, idx.interval_rank,
                                                    idx.interval_starts, idx.interval_lengths, B, C, Z, Y, X, scratch, tile_ws,
                                                    args.tile_voxels, flags, mode)
                    b.record()
                    torch.cuda.synchronize(dev)
                    if it >= 2:
                        ts[mode].append(a.elapsed_time(b))
            for m_, name in names.items():
                floors[name] = sorted(ts[m_])[len(ts[m_]) // 2]
        except _capi.FbbevError:          # another tile / flag set than the default instantiation: no floor reported
            for name in names.values():
                floors.setdefault(name, None)
        del scratch
    P, I = idx.counts.tolist()
    D = cfg.D
    H, W = cfg.feat_hw
    algo_bytes = 4 * B * cfg.n_cams * D * H * W + 4 * B * cfg.n_cams * H * W * C + 4 * (3 * P + 2 * I) + \
        esz * B * Z * Y * X * C
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic, key = None, None
    tj = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            key = f'{cfg.name}_B{B}_tv{args.tile_voxels}' + ('' if args.storage == 'f32' else '_' + args.storage)
            traffic = rec.get(key, {}).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None

    # Extra leg (beside `value`, never as it): the same step with the camera-keyed index cache (the reference's
    # `pre_compute` / `init_acceleration_v2`, view_transformer.py:500-519,607-611, disabled upstream): the six camera tensors are
    # compared on the device every step, the index build returns at once on a hit -- what a deployment with a fixed rig runs
    cached = None
    if world == 1 and rank == 0 and args.storage == 'f32':
        with torch.no_grad():
            vc = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, accelerate=True).to(dev)
            for _ in range(max(3, args.warmup)):
                oc = vc(cam, ctx, depth)
            fence()
            with no_gc():
                tc = time.perf_counter()
                for _ in range(args.steps):
                    oc = vc(cam, ctx, depth)
                fence()
                tc = time.perf_counter() - tc
            ei, ef = prep()                                              # (the fill-rate loop above zeroed `out`: one fresh uncached step)
            pool(ei, ef, out)
            fence()
            cached = {'what': 'forward projection with the camera-keyed index cache hit every step (rank build skipped on the device)',
                      'value': B * args.steps / tc, 'unit': 'samples/s', 'ms_per_step': 1e3 * tc / args.steps,
                      'volume_equals_uncached': bool(torch.equal(oc.permute(0, 1, 4, 2, 3), out))}
            del ei, ef
            del oc, vc

    # Stated baseline beside `value`: the reference's own compiled kernel + its ATen index preparation on this GPU (never the target)
    ref_gpu = None
    if world == 1 and rank == 0 and args.storage == 'f32' and not args.no_reference_gpu:
        try:
            ref_gpu = reference_same_gpu_leg(dev, cfg, B, cam, depth, ctx, vt, idx)
        except Exception as e:
            ref_gpu = {'error': f'{type(e).__name__}: {e}'[:300]}
        torch.cuda.empty_cache()

    # Extra leg (beside `value`, never as it): BASELINE configs[2] -- the backward-projection half of the path on this GPU
    fb = None
    if world == 1 and rank == 0 and not args.no_fb_projection and cfg.name == 'BL2':
        del out
        try:
            fb = fb_projection_leg(dev, args.fb_steps, args.warmup, min(10.0, args.cpu_seconds), with_cpu=not args.no_cpu_baseline)
        except Exception as e:                    # the headline line is printed regardless; the leg reports its own failure
            fb = {'error': f'{type(e).__name__}: {e}'[:300]}

    # Extra leg (beside `value`, never as it): the path's TRAINING step (forward + backward) at the same shapes -- VERDICT r5 item 1
    fbt = None
    if world == 1 and rank == 0 and not args.no_fb_projection and not args.no_fb_train and cfg.name == 'BL2':
        try:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            fbt = fb_projection_train_leg(dev, min(args.fb_steps, 30), args.warmup, min(10.0, args.cpu_seconds), with_cpu=not args.no_cpu_baseline)
        except Exception as e:
            fbt = {'error': f'{type(e).__name__}: {e}'[:300]}

    if rank == 0:
        total = B * world * args.steps
        res = {
            'metric': 'multi-cam samples/sec (forward view transformation: lift + voxel ranking + bev_pool_v2)',
            'value': shard.whole_job_rate(B, args.steps, elapsed, world), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'step_gpu_ms_p10_p50_p90': [pct(0.1), pct(0.5), pct(0.9)],
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.storage, 'accumulate_dtype': 'f32', 'data': 'synthetic',
            'rccl_ranks': ranks, 'rank_devices': devices, 'device': dev_id,
            'launch': ('index build (geometry + ranking + NCHW->NHWC + tile index: ~10 short kernels) replayed from one captured hipGraph '
                       'per step, pooling kernel launched eagerly between the HIP events that time it' if graph is not None else
                       'every kernel launched one by one from the host'),
            'graph_step_equals_eager_step': graph_equal, 'launch_probe': launch_probe, 'graph_capture_error': graph_error,
            'timed_regions': 'cyclic garbage collector paused (collected before, re-enabled after); barrier + device synchronisation on both sides',
            'config': {'workload': f'FB-OCC forward projection, ' + ('BASELINE configs[1] ' if cfg.name == 'BL2' else '') +
                                   f'({cfg.name}): 6x{cfg.input_size[0]}x{cfg.input_size[1]} in, '
                                   f'feat {H}x{W}, D={D}, C={C}, grid {X}x{Y}x{Z}; index tensors rebuilt every step',
                       'samples_per_gpu': B, 'global_batch': B * world, 'points_kept': P, 'intervals': I,
                       'tile_voxels': args.tile_voxels, 'pool_flags': hex(flags), 'volume_storage': args.storage, 'parallelism': f'dp{world} (independent samples, no collective)'},
            'roofline': {'kernel': 'k_pool_fwd_dense2', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': None if traffic is None else f'profiles/pmc_traffic.json[{key}] (rocprofv3 --pmc FETCH_SIZE / '
                                                                       'WRITE_SIZE passes of this command, recorded earlier -- not measured in this run)',
                         'algorithmic_bytes_per_launch': algo_bytes, 'kernel_ms': kern_ms,
                         'device_fill_GBps': fill_gbs, 'frac_of_device_fill': achieved / fill_gbs if fill_gbs > 0 else None,
                         'store_floor_ms': floors.get('store_floor_ms'), 'no_gather_ms': floors.get('no_gather_ms'),
                         'no_store_ms': floors.get('no_store_ms'),
                         'kernel_interleaved_ms': floors.get('kernel_interleaved_ms'),
                         'store_floor_over_kernel': (floors['store_floor_ms'] / floors['kernel_interleaved_ms'])
                         if floors.get('store_floor_ms') and floors.get('kernel_interleaved_ms') else None,
                         'store_floor_what': 'the same k_pool_fwd_dense2 instantiation, grid, tile walk, XCD order and sc1-nt stores with the gathers '
                                             'compiled out (store_floor_ms: stores alone; no_gather_ms: metadata + index staging + LDS tile + stores; no_store_ms: everything but the stores), '
                                             'launched interleaved with the real kernel (K, 1, 2, 3, K, ...), MEDIAN HIP-event time of 15 launches each, after the timed region; store_floor_over_kernel = store_floor_ms / kernel_interleaved_ms'},
        }
        if alt is not None:
            res['bf16_storage'] = alt
        if cached is not None:
            res['index_cache'] = cached
        if fb is not None:
            res['fb_projection'] = fb
        if fbt is not None:
            res['fb_projection_train'] = fbt
        if ref_gpu is not None:
            res['reference_same_gpu'] = ref_gpu
            if ref_gpu.get('samples_per_s'):
                res['reference_same_gpu']['value_over_reference_s2'] = res['value'] / ref_gpu['samples_per_s']
        if piped is not None:
            res['pipelined'] = piped
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg, args.cpu_seconds)
            res['gpu_over_cpu'] = res['value'] / res['cpu_baseline']['value']
            try:
                res['cpu_baseline_c_openmp'] = cpu_baseline_c(cfg, max(3.0, args.cpu_seconds / 3))
            except OSError as e:                  # oracle/_build not built on this box
                res['cpu_baseline_c_openmp'] = {'error': str(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def run_train(args):
    """BASELINE configs[3]: one DDP training step of the whole FB-OCC detector per rank-batch of `--batch` samples
    (default 4 per GPU; 8 GPUs = global batch 32).  Reference: tools/dist_train.sh:10-20 -> mmdet3d/apis/train.py:229-233
    (MMDistributedDataParallel + AdamW lr 2e-4 wd 1e-2, grad clip 5: cfg :360-364), FBOCC.forward_train fbocc.py:400-461.
    Timed step = zero grads -> forward_train -> backward (bucket all-reduces launched from autograd hooks, overlapping
    it) -> wait -> clip -> AdamW.  Nothing is skipped or cached; history fusion runs with its 16-frame state."""
    import torch
    import torch.distributed as dist
    from fb_bev_amd import shard, synthetic as S
    from fb_bev_amd.fbocc import FBOCC
    world, rank, local_rank = shard.world()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    shard.init('nccl', dev)
    ranks, devices = check_ranks(args, world, rank, dev)
    B = args.batch if args.batch != 16 else 4          # 16 is the forward mode's default; configs[3] is 4 per GPU
    from fb_bev_amd import configs
    pc = S.CONFIGS['REF']
    seed = shard.shard_seed(rank)
    g = torch.Generator().manual_seed(seed)
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=seed, bda_aug=True)]
    img = torch.randn(B, 6, 3, 256, 704, generator=g).to(dev)
    gt_depth = torch.rand(B, 6, 256, 704, generator=g) * 40 + 2
    gt_depth[torch.rand(gt_depth.shape, generator=g) > 0.03] = 0          # ~3 % LiDAR returns
    gt_occ = torch.randint(1, 19, (B, 200, 200, 16), generator=g)
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.6] = 18
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.4] = 255
    gt_occ, gt_depth = gt_occ.to(dev), gt_depth.to(dev)
    ego = torch.eye(4)
    ego[0, 3] = 0.5
    img_inputs = [img] + cam

    def metas(first):
        return [dict(sequence_group_idx=rank * B + b, start_of_sequence=first, curr_to_prev_ego_rt=ego, index=b) for b in range(B)]

    def build(conv_dtype):
        cfg = configs.model_block(configs.SHIPPED)          # package data: the shipped config's unchanged `model` block
        cfg.pop('type')
        ex = dict(with_cp=False, mfma_conv3d_train=(args.conv == 'mfma'))
        if conv_dtype == 'bf16':
            ex.update(img_dtype='bf16', depth_dtype='bf16')
        torch.manual_seed(0)                               # identical initial parameters on every rank
        model = FBOCC(**cfg, execution=ex).to(dev).train()
        model, buckets = shard.prepare_ddp(model, sync_bn=args.sync_bn, bucket_bytes=args.bucket_mb << 20)
        opt = torch.optim.AdamW(buckets.params, lr=2e-4, weight_decay=1e-2)
        return model, buckets, opt

    def make_step(model, buckets, opt, ar_ev):
        params = buckets.params

        def step(i=None, first=False):
            buckets.zero_grad()
            losses = model(return_loss=True, img_inputs=img_inputs, img_metas=metas(first), gt_occupancy=gt_occ, gt_depth=gt_depth)
            total = model.parse_losses(losses)
            total.backward()                              # hooks launch each bucket's all-reduce as soon as it is complete
            if i is not None:
                ar_ev[i][0].record()
            buckets.finish()                              # exposed (non-overlapped) tail of the gradient all-reduce
            if i is not None:
                ar_ev[i][1].record()
            torch.nn.utils.clip_grad_norm_(params, max_norm=5, norm_type=2)
            opt.step()
            return total
        return step

    def timed(step, steps, warmup):
        total = step(first=True)
        for _ in range(max(0, warmup - 1)):
            total = step()
        shard.fence(dev)
        with no_gc():
            t0 = time.perf_counter()
            for i in range(steps):
                total = step(i)
            shard.fence(dev)
            dt = time.perf_counter() - t0
        return shard.max_over_ranks(dt, dev), total

    model, buckets, opt = build(args.conv_dtype)
    params = buckets.params
    ar_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    step = make_step(model, buckets, opt, ar_ev)
    elapsed, total = timed(step, args.steps, args.warmup)
    ar_tail_ms = sum(a.elapsed_time(b) for a, b in ar_ev) / max(1, args.steps)
    n_params, grad_bytes, n_buckets = sum(p.numel() for p in params), buckets.nbytes, len(buckets.buckets)
    # the OTHER arithmetic of the 2-D stacks beside it (VERDICT r3/r4 hygiene: the scope table's S5 is fp32, this line's default
    # bf16): the same step on a second model, half the steps; reported as `other_conv_dtype`, never as `value`
    other = None
    if not args.no_alt_dtype:
        od = 'f32' if args.conv_dtype == 'bf16' else 'bf16'
        if not os.environ.get('FBBEV_TRAIN_PROFILE'):
            del model, opt, step
            torch.cuda.empty_cache()
        m2, b2, o2 = build(od)
        k2 = max(2, args.steps // 2)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k2)]
        el2, _ = timed(make_step(m2, b2, o2, ev2), k2, max(2, args.warmup // 2))
        other = {'conv_dtype': od, 'value': shard.whole_job_rate(B, k2, el2, world), 'unit': 'samples/s', 'ms_per_step': 1e3 * el2 / k2, 'steps': k2}
        del m2, b2, o2
    if rank == 0 and os.environ.get('FBBEV_TRAIN_PROFILE'):
        # diagnostics only (after the timed region): per-kernel GPU time of two STEADY-STATE steps -- a rocprofv3 run of
        # the whole command is dominated by the vendor library's solver search in the first step
        from torch.profiler import ProfilerActivity, profile
        if os.environ.get('FBBEV_TRAIN_PROFILE_COPIES'):
            # where the layout copies come from: device time of aten::copy_ by Python call site (one step)
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
                step()
                torch.cuda.synchronize(dev)
            rows = []
            for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
                if e.key in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::_to_copy') and e.device_time_total > 200:
                    rows.append({'op': e.key, 'ms': e.device_time_total / 1e3, 'calls': e.count, 'shapes': str(e.input_shapes)[:160],
                                 'stack': [fr for fr in e.stack if 'fb_bev_amd' in fr or 'bench.py' in fr][:5]})
            rows.sort(key=lambda r: -r['ms'])
            json.dump(rows[:60], open(os.environ['FBBEV_TRAIN_PROFILE'] + '.copies.json', 'w'), indent=1)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(2):
                step()
            torch.cuda.synchronize(dev)
        evs = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
        tot = sum(e.device_time_total for e in evs)
        json.dump({'steps': 2, 'gpu_ms_per_step': tot / 2e3, 'ms_per_step_wall': 1e3 * elapsed / args.steps,
                   'kernels': [{'name': e.key[:120], 'calls_per_step': e.count / 2, 'ms_per_step': e.device_time_total / 2e3,
                                'pct': 100.0 * e.device_time_total / tot} for e in evs[:60]]},
                  open(os.environ['FBBEV_TRAIN_PROFILE'], 'w'), indent=1)
    if rank == 0:
        print(json.dumps({
            'metric': 'multi-cam samples/sec (FB-OCC R50 training step: forward_train + backward + gradient all-reduce + AdamW)',
            'value': shard.whole_job_rate(B, args.steps, elapsed, world), 'unit': 'samples/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.conv_dtype == 'f32' else 'bf16 (2-D convolution stacks: image encoder + depth net, fp32 master weights; view transformation, 3-D stacks, losses, gradients: f32)',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[3]: FB-OCC R50 (fbocc-r50-cbgs_depth_16f_16x4_20e) training step, 6x256x704 in, '
                                   'D=80, 100x100x8 grid, 16-frame history, occupancy + depth losses',
                       'samples_per_gpu': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'parameters': n_params, 'gradient_bytes': grad_bytes,
                       'gradient_buckets': n_buckets, 'bucket_mb': args.bucket_mb, 'sync_bn': bool(args.sync_bn and world > 1),
                       'conv3d_route': args.conv, 'optimizer': 'AdamW lr 2e-4 wd 1e-2, clip 5'},
            'rccl_ranks': ranks, 'rank_devices': devices,
            'allreduce_exposed_ms': ar_tail_ms, 'loss': float(total.detach()), 'conv_dtype': args.conv_dtype, 'other_conv_dtype': other,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
