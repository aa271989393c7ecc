#!/usr/bin/env python3
"""Diagnostic: hipGraph capture / replay of the view transformation, step by step (prints before every device sync)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import synthetic as S
from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D


def say(*a):
    print(*a, flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
    dev = torch.device('cuda:0'); cfg = S.CONFIGS['SMALL']
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    cam = S.camera_rig(cfg, 2, seed=0, bda_aug=True); depth, ctx = S.depth_and_context(cfg, 2, seed=0)
    cam2 = S.camera_rig(cfg, 2, seed=7, bda_aug=True); depth2, ctx2 = S.depth_and_context(cfg, 2, seed=7)
    with torch.no_grad():
        for tag, (c, d, x) in {'seed0': (cam, depth, ctx), 'seed7': (cam2, depth2, ctx2)}.items():
            o = vt([t.to(dev) for t in c], x.to(dev), d.to(dev)); torch.cuda.synchronize()
            say('eager', tag, float(o.sum()))
        if mode == 'eager':
            return
        cam_s = [t.to(dev).clone() for t in cam]; d_s, c_s = depth.to(dev).clone(), ctx.to(dev).clone()
        names = ('ranks_bev', 'ranks_depth', 'ranks_feat', 'interval_starts', 'interval_lengths', 'interval_rank')

        def cmp(tag, got, exp):
            P, I = exp.counts.tolist()
            res = {'counts': got.counts.tolist() == [P, I]}
            for nme in names:
                k = P if nme.startswith('ranks') else I
                res[nme] = bool(torch.equal(getattr(got, nme)[:k], getattr(exp, nme)[:k]))
            say(tag, res)
        if mode == 'index':                     # capture ONLY the rank build
            vt.build_index_from_cams(*cam_s)
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                vt.build_index_from_cams(*cam_s)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gi = vt.build_index_from_cams(*cam_s)
            g.replay(); torch.cuda.synchronize()
            cmp('index replay same', gi, vt.build_index_from_cams(*cam_s))
            for dst, src in zip(cam_s, cam2):
                dst.copy_(src)
            torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            cmp('index replay new ', gi, vt.build_index_from_cams(*cam_s))
            g.replay(); torch.cuda.synchronize()
            cmp('index replay new2', gi, vt.build_index_from_cams(*cam_s))
            return
        e0 = vt(cam_s, c_s, d_s).clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            vt(cam_s, c_s, d_s)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = vt(cam_s, c_s, d_s)
        for i in range(2):
            g.replay(); torch.cuda.synchronize(); say('replay', i, 'same data ok', bool(torch.equal(out, e0)))
        for dst, src in zip(cam_s, cam2):
            dst.copy_(src)
        d_s.copy_(depth2); c_s.copy_(ctx2); torch.cuda.synchronize(); say('inputs changed')
        g.replay(); torch.cuda.synchronize()
        e1 = vt([t.to(dev) for t in cam2], ctx2.to(dev), depth2.to(dev))
        e_mixed = vt([t.to(dev) for t in cam], ctx2.to(dev), depth2.to(dev))      # OLD cameras, new features
        say('replay new data: equals eager(new)', bool(torch.equal(out, e1)), '| equals old result', bool(torch.equal(out, e0)),
            '| equals eager(old cams, new feats)', bool(torch.equal(out, e_mixed)), '| mismatching voxels vs eager(new)',
            int((out != e1).any(1).sum()))
        g.replay(); torch.cuda.synchronize()
        say('second replay new data: equals eager(new)', bool(torch.equal(out, e1)))


if __name__ == '__main__':
    main()
