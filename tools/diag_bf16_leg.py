import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
dev = torch.device('cuda:0'); cfg = S.CONFIGS['BL2']; B = 16
cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
depth, ctx = (t.to(dev) for t in S.depth_and_context(cfg, B, seed=0))
C = cfg.channels
def leg(dt, label):
    v = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample, out_dtype=dt).to(dev)
    tv, fl = v.tiling(cfg.n_cams); Z, Y, X = v.grid_zyx
    tws = v._tile_ws(dev, B, tv)
    out = torch.empty((B, C, Z, Y, X), dtype=dt, device=dev)
    def step(sync=False):
        ts = [time.perf_counter()]
        ix = v.build_index_from_cams(*cam)
        if sync: torch.cuda.synchronize()
        ts.append(time.perf_counter())
        ft = _capi.nchw_to_nhwc(ctx)
        if sync: torch.cuda.synchronize()
        ts.append(time.perf_counter())
        _capi.pool_tile_index(ix.interval_rank, ix.interval_starts, ix.counts, ix.n, B, Z, Y, X, tws, tv)
        if sync: torch.cuda.synchronize()
        ts.append(time.perf_counter())
        _capi.bev_pool_v2_dense_fwd(depth, ft, ix.ranks_depth, ix.ranks_feat, ix.interval_rank, ix.interval_starts, ix.interval_lengths, B, C, Z, Y, X, out, tws, tv, fl)
        if sync: torch.cuda.synchronize()
        ts.append(time.perf_counter())
        return [1e3 * (b - a) for a, b in zip(ts, ts[1:])]
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print(label, 'ms/step back-to-back', round(1e3 * (time.perf_counter() - t0) / 30, 4), 'tile', tv, hex(fl))
    acc = [0, 0, 0, 0]
    for _ in range(10):
        for i, x in enumerate(step(True)): acc[i] += x / 10
    print(label, 'synced parts (rank, nchw, tile, pool) ms:', [round(a, 4) for a in acc])
    t0 = time.perf_counter()
    for _ in range(30): step()
    host = 1e3 * (time.perf_counter() - t0) / 30
    torch.cuda.synchronize()
    print(label, 'host issue ms/step', round(host, 4))
leg(torch.float32, 'f32')
leg(torch.bfloat16, 'bf16')
leg(torch.float32, 'f32 again')
