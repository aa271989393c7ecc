"""dense bev_pool_v2 at BASELINE configs[2] B = 4 in isolation: plain, with the re-add epilogue, and each behind a cache flush
(the S3 trace shows the re-add form at 268 us against 141 us for the plain one).  GPU only: python tools/dbg_pool_add.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import configs, synthetic as S
from fb_bev_amd.fb_view_transform import FBViewTransform
dev = torch.device('cuda:0')
pc = S.CONFIGS['BL2']; B = 4
X, Y, Z = pc.grid_xyz
gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                        grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample, num_levels=1)
m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(dev).eval()
fp = m.forward_projection
cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
depth, ctx = (t.to(dev) for t in S.depth_and_context(pc, B, seed=0))
with torch.no_grad():
    parts = fp.pooling_inputs(cam, ctx, depth)
    addend = torch.randn(B, pc.channels, Y, X, device=dev)
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB: past the L2s and the 256 MB memory-side cache
    def timed(fn, flush, n=12):
        ts = []
        for _ in range(n):
            if flush:
                big.fill_(1.0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 1)
    for tag, fn in (('plain', lambda: fp.pooled_volume(parts)), ('re-add', lambda: fp.pooled_volume(parts, addend=addend))):
        for flush in (False, True):
            print(tag, 'after a 1 GiB fill' if flush else 'back to back', timed(fn, flush), 'us (incl. torch.empty of the volume)', flush=True)
