#!/usr/bin/env python3
"""The whole built path on one GPU, from image features to the temporally fused BEV volume, built from the shipped
detector config (committed extraction): CM_DepthNet -> FBViewTransform (lift-splat, backward projection, re-add) ->
TemporalHistoryFusion, over a synthetic driving sequence.   python tools/time_path.py [B] [bf16]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from fb_bev_amd import config as C, synthetic as S


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
    dev = torch.device('cuda:0')
    from fb_bev_amd import configs
    model = C.path_blocks(configs.model_block())
    torch.manual_seed(0)
    depth_net = C.build_depth_net(model, compute_dtype=torch.bfloat16 if bf16 else torch.float32).to(dev).eval()
    fvt, hist = C.build_view_transformation(model)
    fvt, hist = fvt.to(dev).eval(), hist.to(dev).eval()
    hist.do_history = True
    cfg = S.CONFIGS['REF']
    cam = [t.to(dev) for t in S.camera_rig(cfg, B, seed=0, bda_aug=False)]
    feats = [torch.randn(B, 6, 256, 16, 44, device=dev) for _ in range(4)]          # image-neck output per frame
    ego = torch.eye(4); ego[0, 3] = 0.8

    def frame(i, first=False):
        mlp = depth_net.get_mlp_input(*cam)
        context, depth = depth_net(feats[i % 4], mlp)
        bev = fvt(cam, context, depth)
        metas = [dict(sequence_group_idx=b, start_of_sequence=first, curr_to_prev_ego_rt=ego) for b in range(B)]
        return hist.fuse_history(bev, metas, cam[5])

    def ev_time(fn, n=10):
        ts = []
        for i in range(n):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(i); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        return sorted(ts)[len(ts) // 2]

    with torch.no_grad():
        frame(0, first=True)
        t0 = time.perf_counter()
        i = 1
        while time.perf_counter() - t0 < 2.0:          # sustained load first: the engine clock needs ~1 s to ramp up,
            out = frame(i)                              # and the MFMA-bound kernels scale with it (3x between cold and warm)
            i += 1
        torch.cuda.synchronize()
        t_all = ev_time(lambda i: frame(i))
        mlp = depth_net.get_mlp_input(*cam)
        t_dn = ev_time(lambda i: depth_net(feats[i % 4], mlp))
        context, depth = depth_net(feats[0], mlp)
        t_fb = ev_time(lambda i: fvt(cam, context, depth))
        bev = fvt(cam, context, depth)
        metas = [dict(sequence_group_idx=b, start_of_sequence=False, curr_to_prev_ego_rt=ego) for b in range(B)]
        t_h = ev_time(lambda i: hist.fuse_history(bev, metas, cam[5]))
    print(json.dumps({'B': B, 'depth_net_dtype': 'bf16' if bf16 else 'f32', 'out': list(out.shape), 'ms_frame': round(t_all, 3),
                      'ms_depth_net': round(t_dn, 3), 'ms_forward_backward_projection': round(t_fb, 3),
                      'ms_history_fusion': round(t_h, 3), 'frames_per_s': round(1e3 * B / t_all, 1)}))


if __name__ == '__main__':
    main()
