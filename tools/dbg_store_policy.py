"""What do a kernel's stores leave behind for the NEXT kernel's store stream?  The dense pooling kernel (819 MB of sc1-nt stores at
BASELINE configs[2], B = 4) takes 141 us alone and 206 us at the end of the S3 step.  Here it is timed right behind a fill of N MB
written with each store policy of fbbev_store4.  GPU only: python tools/dbg_store_policy.py"""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import _capi, synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
dev = torch.device('cuda:0')
pc = S.CONFIGS['BL2']; B = 4
fp = LSSViewTransformerFunction3D(pc.grid_config, pc.input_size, pc.downsample).to(dev)
cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
depth, ctx = (t.to(dev) for t in S.depth_and_context(pc, B, seed=0))
X, Y, Z = pc.grid_xyz
names = ['plain', 'nt', 'sc1', 'sc0 sc1', 'sc1 nt', 'sc0 nt', 'sc0 sc1 nt', 'sc0']
with torch.no_grad():
    parts = fp.pooling_inputs(cam, ctx, depth)
    addend = torch.randn(B, pc.channels, Y, X, device=dev)
    for mb in (128, 512):
        big = torch.empty(mb << 18, dtype=torch.float32, device=dev)
        for pol in [-1] + list(range(8)):
            ts, tf = [], []
            for it in range(10):
                fp.pooled_volume(parts, addend=addend)                      # steady state in front of the fill
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                a.record()
                if pol >= 0:
                    _capi._check(_capi.lib().fbbev_diag_fill(big.data_ptr(), big.numel(), pol, _capi._stream()), 'fill')
                b.record()
                fp.pooled_volume(parts, addend=addend)
                c.record()
                torch.cuda.synchronize()
                if it >= 2:
                    tf.append(a.elapsed_time(b) * 1e3); ts.append(b.elapsed_time(c) * 1e3)
            ts.sort(); tf.sort()
            print(json.dumps({'fill_MB': mb, 'policy': 'none' if pol < 0 else names[pol], 'fill_us': round(tf[len(tf) // 2], 1),
                              'pool_after_us': round(ts[len(ts) // 2], 1)}), flush=True)
