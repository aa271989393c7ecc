import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from exp_pool2 import per_launch
dev = torch.device('cuda:0'); cfg = S.CONFIGS['BL2']; B = 16
cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=True)]
depth, ctx = S.depth_and_context(cfg, B, seed=0); depth, ctx = depth.to(dev), ctx.to(dev)
vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
Z, Y, X = vt.grid_zyx; C = cfg.channels
idx = vt.build_index(vt.get_lidar_coor(*cam)); feat = ctx.permute(0, 1, 3, 4, 2).contiguous()
ws = torch.empty(_capi.pool_dense_workspace_bytes(B, Z, Y, X), dtype=torch.uint8, device=dev)
zero = torch.zeros_like(idx.counts)
ZYX = Z * Y * X
def padded(pad_c, pad_b=0):
    sc = ZYX + pad_c; sb = C * sc + pad_b
    buf = torch.empty(B * sb + 64, device=dev)
    return torch.as_strided(buf, (B, C, Z, Y, X), (sb, sc, Y * X, X, 1))
CL = 0x100000
out_cl = torch.empty((B, Z, Y, X, C), device=dev)
from fb_bev_amd.bev_pool import bev_pool_v2
rb, rd_, rf_, st_, ln_ = idx.exact()
ref = None
for tv in (8, 16, 32, 64):
    for st in (0, 4):
        for lg in (None, 0, 2, 4, 6):
            base = _capi.pool_flags(store=st, csplit=1, wg=256, swizzle=lg is not None, swz_log2=lg or 0, cpl8=False) | CL
            rec = {'variant': f'CL_tv{tv}_st{st}_swz{lg}'}
            try:
                for name, counts in (('full', idx.counts), ('empty', zero)):
                    ti = lambda: _capi.pool_tile_index(idx.interval_rank, idx.interval_starts, counts, idx.n, B, Z, Y, X, ws, tv, CL)
                    ti()
                    f = lambda: _capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts,
                                                            idx.interval_lengths, B, C, Z, Y, X, out_cl, ws, tv, base)
                    rec[name + '_ms'] = round(per_launch(None, f, iters=8, warm=2), 4)
                    if name == 'full':
                        if ref is None:
                            ref = out_cl.clone()
                        rec['bits_ok'] = bool(torch.equal(out_cl, ref))
                        rec['tile_index_ms'] = round(per_launch(None, ti, iters=8, warm=2), 4)
            except Exception as e:
                rec['error'] = str(e)
            print(json.dumps(rec), flush=True)
# cross-check the channels-last result against the (B,C,Z,Y,X) kernel
out = torch.empty((B, C, Z, Y, X), device=dev)
_capi.pool_tile_index(idx.interval_rank, idx.interval_starts, idx.counts, idx.n, B, Z, Y, X, ws, 128)
_capi.bev_pool_v2_dense_fwd(depth, feat, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts, idx.interval_lengths,
                            B, C, Z, Y, X, out, ws, 128, _capi.DEFAULT_POOL_FLAGS)
print(json.dumps({'variant': 'CL_equals_BCZYX', 'ok': bool(torch.equal(ref.permute(0, 4, 1, 2, 3), out))}))
print(json.dumps({'variant': 'torch_zero_', 'ms': round(per_launch(None, lambda: out.zero_()), 4)}))
