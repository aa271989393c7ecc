#!/usr/bin/env python3
"""Time the forward+backward view transformation (BASELINE configs[2] scope) on one GPU."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd import configs, synthetic as S
from fb_bev_amd.fb_view_transform import FBViewTransform

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'REF'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    levels = int(sys.argv[4]) if len(sys.argv) > 4 else 1        # BASELINE configs[2]: 4 attention levels
    vdt = {'f32': None, 'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[5] if len(sys.argv) > 5 else 'f32']   # camera-token storage
    dev = torch.device('cuda:0')
    pc = S.CONFIGS[name]
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample,
                            num_levels=levels)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(dev).eval()
    if vdt is not None:
        from fb_bev_amd.backward_projection import DA_SpatialCrossAttention
        for mod in m.modules():
            if isinstance(mod, DA_SpatialCrossAttention):
                mod.value_dtype = vdt
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(pc, B, seed=0)
    depth, ctx = depth.to(dev), ctx.to(dev)
    mlvl = None
    if levels > 1:      # synthetic pyramid: 1x (= the depth net's level, must be level 0), 2x, 1/2x, 1/4x
        H, W = ctx.shape[-2:]
        g = torch.Generator().manual_seed(5)
        shapes = [(H, W), (2 * H, 2 * W), (H // 2, W // 2), (H // 4, W // 4)][:levels]
        mlvl = [torch.randn(B, pc.n_cams, pc.channels, h, w, generator=g).to(dev) for h, w in shapes]
        mlvl[0] = ctx
    with torch.no_grad():
        for _ in range(3):
            out = m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fp = m.forward_projection
        t0 = time.perf_counter()
        for _ in range(steps):
            b = fp(cam, ctx, depth)
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t0) / steps
        # the same two scopes replayed from captured hipGraphs (no host launch overhead)
        res = {}
        for tag, fn in (('fb', lambda: m(cam, ctx, depth, mlvl_feats=mlvl)), ('forward_only', lambda: fp(cam, ctx, depth))):
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fn()
            torch.cuda.current_stream().wait_stream(st)
            with torch.cuda.graph(g):
                o = fn()
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            res['ms_' + tag + '_graph'] = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({'config': name, 'B': B, 'levels': levels, 'camera_tokens': str(vdt or 'f32'), 'bev': [Y, X], 'out': list(out.shape), 'ms_fb': dt * 1e3, 'ms_forward_only': dt_f * 1e3,
                      'samples_per_s_fb': B / dt, **res, 'samples_per_s_fb_graph': B / (res['ms_fb_graph'] * 1e-3)}))

if __name__ == '__main__':
    main()
