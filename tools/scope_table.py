#!/usr/bin/env python3
"""Scopes S1-S5 of SURVEY 8d on one MI355X, from ONE script: every DESIGN section-5 number is reproducible from the
JSON this writes (profiles/rNN_scope_table.json).  fp32, synthetic data, p10 / p50 / p90 of per-iteration GPU time
(HIP events around each iteration, device synchronised in between).

    python tools/scope_table.py [out.json] [quick]

  S1  bev_pool_v2 op only (dense fused kernel, indices given)            REF B=16, BL2 B=16
  S2  forward projection: geometry + ranking + tile index + pooling        REF B=16, BL2 B=16   (the bench metric)
  S2c the same with the camera-keyed index cache hit (accelerate=True)     REF B=16, BL2 B=16
  S3  S2 + backward projection + re-add (FBViewTransform)                  REF B=1 / B=4 (1 level), BL2 grid B=4 (4 levels = BASELINE configs[2])
  S4  full detector, images -> occupancy ids (shipped config)              B=1: fp32-MFMA convolution route (default), bf16_tiled route
  S5  full training step (forward_train + backward + clip + AdamW)         B=2, B=4 (= BASELINE configs[3] per-GPU batch)
"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, configs, shard, synthetic as S
from fb_bev_amd.fb_view_transform import FBViewTransform
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D

DEV = torch.device('cuda:0')
ROWS = []


def pct(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return [round(ts[int(q * (len(ts) - 1))], 4) for q in (0.1, 0.5, 0.9)]


def row(scope, config, B, ms, **extra):
    r = dict(scope=scope, config=config, B=B, ms_p10_p50_p90=ms, samples_per_s=round(1e3 * B / ms[1], 2), **extra)
    ROWS.append(r)
    print(json.dumps(r), flush=True)


def s1_s2(name, B, n):
    cfg = S.CONFIGS[name]
    cam = [t.to(DEV) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
    depth, ctx = (t.to(DEV) for t in S.depth_and_context(cfg, B, seed=0))
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(DEV)
    Z, Y, X = vt.grid_zyx; C = cfg.channels
    idx = vt.build_index_from_cams(*cam); feat = _capi.nchw_to_nhwc(ctx)
    tv, fl = vt.tiling(cfg.n_cams)
    ws = vt._tile_ws(DEV, B, tv)
    out = torch.empty((B, C, Z, Y, X), device=DEV)
    _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, tv)
    P, I = idx.counts.tolist(); H, W = cfg.feat_hw
    algo = 4 * B * cfg.n_cams * cfg.D * H * W + 4 * B * cfg.n_cams * H * W * C + 4 * (3 * P + 2 * I) + out.numel() * 4
    ms = pct(lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                                 idx.interval_lengths, B, C, Z, Y, X, out, ws, tv, fl), n)
    row('S1 bev_pool_v2 dense forward', name, B, ms, algorithmic_bytes=algo, frac_of_8TBs=round(algo / ms[1] / 1e6 / 8000, 3))
    with torch.no_grad():
        row('S2 forward projection (indices rebuilt every call)', name, B, pct(lambda: vt(cam, ctx, depth), n))
        row('S2 rank build only (fbbev_lift_rank_build)', name, B, pct(lambda: vt.build_index_from_cams(*cam), n))
        vc = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, accelerate=True).to(DEV)
        row('S2c forward projection, camera-keyed index cache hit', name, B, pct(lambda: vc(cam, ctx, depth), n))
        vh = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, out_dtype=torch.bfloat16).to(DEV)
        row('S2 forward projection, bf16 volume storage (fp32 in-order sums rounded once)', name, B, pct(lambda: vh(cam, ctx, depth), n))


def s3(name, B, levels, n):
    pc = S.CONFIGS[name]
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample,
                            num_levels=levels)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(DEV).eval()
    cam = [t.to(DEV) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
    depth, ctx = (t.to(DEV) for t in S.depth_and_context(pc, B, seed=0))
    mlvl = None
    if levels > 1:
        H, W = ctx.shape[-2:]
        g = torch.Generator().manual_seed(5)
        shapes = [(H, W), (2 * H, 2 * W), (H // 2, W // 2), (H // 4, W // 4)][:levels]
        mlvl = [torch.randn(B, pc.n_cams, pc.channels, h, w, generator=g).to(DEV) for h, w in shapes]
        mlvl[0] = ctx
    with torch.no_grad():
        row(f'S3 forward + backward projection + re-add ({levels} attention level{"s" if levels > 1 else ""})', name, B,
            pct(lambda: m(cam, ctx, depth, mlvl_feats=mlvl), n))
        from fb_bev_amd.graphed import Graphed
        g = Graphed(m, cam, ctx, depth, mlvl_feats=mlvl)
        row(f'S3g the same call replayed from a captured hipGraph (fb_bev_amd.graphed.Graphed)', name, B,
            pct(lambda: g(cam, ctx, depth, mlvl_feats=mlvl), n))
        if levels > 1:
            from fb_bev_amd.backward_projection import DA_SpatialCrossAttention
            for mod in m.modules():
                if isinstance(mod, DA_SpatialCrossAttention):
                    mod.value_dtype = torch.bfloat16
            row(f'S3 the same with bf16 camera tokens in the cross-attention (fp32 accumulate)', name, B,
                pct(lambda: m(cam, ctx, depth, mlvl_feats=mlvl), n))
            return
    if B == 4 and levels == 1:        # training step of the path alone: forward + backward of FBViewTransform
        m.train()
        dg, cg = depth.clone().requires_grad_(), ctx.clone().requires_grad_()
        w = torch.randn(B, pc.channels, Y, X, Z, device=DEV)

        def step():
            for p_ in m.parameters():
                p_.grad = None
            dg.grad = cg.grad = None
            (m(cam, cg, dg) * w).sum().backward()
        row('S3t training step of the path (forward + backward of FBViewTransform, fused DA backward)', name, B, pct(step, n))


def s6(n):
    """BASELINE configs[4] as ONE path: forward projection + backward projection + re-add + 16-frame temporal fusion with
    the history ring in fp16, 6x512x1408 input (feat 32x88, D=118), 400x400x16 grid, one sample per GPU."""
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    pc = S.CONFIGS['BL5']
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(DEV).eval()
    dx = [pc.grid_config[a][2] for a in 'xyz']
    bx = [pc.grid_config[a][0] + pc.grid_config[a][2] / 2 for a in 'xyz']
    hist = TemporalHistoryFusion(dx, bx, single_bev_num_channels=pc.channels, history_cat_num=16,
                                 history_dtype=torch.float16).to(DEV).eval()
    hist.do_history = True
    cam = [t.to(DEV) for t in S.camera_rig(pc, 1, seed=0, bda_aug=False)]
    depth, ctx = (t.to(DEV) for t in S.depth_and_context(pc, 1, seed=0))
    ego = torch.eye(4); ego[0, 3] = 0.8
    state = {'first': True}

    def frame():
        bev = m(cam, ctx, depth)
        out = hist.fuse_history(bev, [dict(sequence_group_idx=0, start_of_sequence=state['first'], curr_to_prev_ego_rt=ego)], cam[5])
        state['first'] = False
        return out
    with torch.no_grad():
        vt_ms = None
        for comp, lay, label in ((torch.float32, 'planar', 'fp32 MFMA'), (torch.float32, 'voxel_major', 'fp32 MFMA'),
                                 (torch.bfloat16, 'planar', 'bf16 MFMA (fp32 accumulate)'),
                                 ('bf16x3', 'voxel_major', 'bf16 MFMA x3 (split operands: fp32-grade)'),
                                 (torch.bfloat16, 'voxel_major', 'bf16 MFMA (fp32 accumulate)')):
            hist.history_compute, hist.ring_layout = comp, lay
            hist.reset(); state['first'] = True
            frame()
            ms = pct(frame, n, warm=2)
            vt_ms = vt_ms or pct(lambda: m(cam, ctx, depth), n, warm=1)
            row('S6 BASELINE configs[4] path: lift-splat + backward projection + re-add + 16-frame history (fp16 ring)',
                'BL5 (400x400x16, 6x512x1408)', 1, ms, history_convs=label, ring_layout=lay, view_transformation_ms_p50=vt_ms[1],
                history_ring_GB=round(hist.history_bev.numel() * 2 / 2 ** 30, 1),
                peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
    del m, hist
    torch.cuda.empty_cache()


def s4_s5(quick):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import time_full as T
    for prec in (True, 'bf16_tiled'):
        m = T.build('f32', mfma=prec).to(DEV).eval()
        m.do_history = True
        img_inputs, metas, _, _ = T.inputs(1, DEV)
        with torch.no_grad():
            m.predict_occupancy(img_inputs, metas(True))
            row('S4 full detector forward, images -> occupancy ids', 'shipped fbocc-r50 (REF grid)', 1,
                pct(lambda: m.predict_occupancy(img_inputs, metas(False)), 6 if quick else 15),
                conv_route='fp32 MFMA (default)' if prec is True else 'bf16 MFMA, LDS-tiled 3x3x3')
        del m
        torch.cuda.empty_cache()
    for B in ((2,) if quick else (2, 4)):
        m = T.build('f32', mfma_train=True).to(DEV).train()
        img_inputs, metas, gt_occ, gt_depth = T.inputs(B, DEV)
        m, buckets = shard.prepare_ddp(m, sync_bn=False)
        params = buckets.params
        opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-2)
        state = {'first': True}

        def step():
            buckets.zero_grad()
            losses = m(return_loss=True, img_inputs=img_inputs, img_metas=metas(state['first']), gt_occupancy=gt_occ, gt_depth=gt_depth)
            state['first'] = False
            m.parse_losses(losses).backward()
            buckets.finish()
            torch.nn.utils.clip_grad_norm_(params, max_norm=5, norm_type=2)
            opt.step()
        row('S5 full training step (forward_train + backward + clip + AdamW)', 'shipped fbocc-r50 (REF grid)', B,
            pct(step, 3 if quick else 5, warm=2), conv3d_route='fbbev_conv3d_* (fwd + dgrad + wgrad)',
            note='2-D stacks on the vendor library in fp32: its naive fallback kernels dominate (profiles/r02_rocprofv3_train_step.csv)',
            peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
        del m, opt, buckets
        torch.cuda.empty_cache()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.json') else os.path.join(ROOT, 'gpurun_out', 'scope_table.json')
    quick = 'quick' in sys.argv
    n = 20 if quick else 100                 # at least 100 timed iterations (round-2 tables were taken with 50)
    only6 = 'only_s6' in sys.argv
    for name in (() if only6 else ('REF', 'BL2')):
        s1_s2(name, 16, n)
    if not only6:
        s3('REF', 1, 1, n); s3('REF', 4, 1, n); s3('BL2', 4, 4, n)
    if 'only_s6' in sys.argv:
        ROWS.clear()
    s6(8 if quick else 15)
    if 'only_s6' not in sys.argv:
        s4_s5(quick)
    json.dump({'device': torch.cuda.get_device_name(0), 'torch': torch.__version__, 'rows': ROWS}, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
