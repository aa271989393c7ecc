#!/usr/bin/env python3
"""Scopes S4 / S5 of SURVEY 8d on one GPU: the whole FB-OCC detector built from the shipped config block
(fb_bev_amd/data/fbocc_model_blocks.json, extracted from occupancy_configs/fb_occ/fbocc-r50-cbgs_depth_16f_16x4_20e.py)
with random-init weights and synthetic inputs of SURVEY Appendix B.

    python tools/time_full.py infer B [f32|bf16] [mfma]   S4: images -> occupancy class ids (device), per-stage split;
                                                          mfma | mfma_bf16 = conv stacks on fbbev_conv3d_ndhwc (fp32 MFMA) / _bf16
    python tools/time_full.py train B [f32|bf16] [mfma]   S5: forward_train + backward + grad all-reduce + clip + AdamW step

bf16 = convolution stacks (image encoder, depth net, voxel encoder, head) under bf16 autocast; the view transformation,
history fusion and losses stay fp32.  A trailing `tune` turns on the vendor library's algorithm search
(torch.backends.cudnn.benchmark).  Prints one JSON line.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fb_bev_amd import shard, synthetic as S  # noqa: E402
from fb_bev_amd.fbocc import FBOCC  # noqa: E402


def build(dtype, with_cp=False, mfma=False, mfma_train=False):
    from fb_bev_amd import configs
    cfg = configs.model_block()
    cfg.pop('type')
    ex = dict(with_cp=with_cp, mfma_conv3d=mfma, mfma_conv3d_train=mfma_train)     # mfma: False | True | 'bf16'
    if dtype == 'bf16':
        ex.update(img_dtype='bf16', depth_dtype='bf16', voxel_dtype='bf16', head_dtype='bf16')
    torch.manual_seed(0)
    return FBOCC(**cfg, execution=ex)


def inputs(B, dev, seed=0):
    pc = S.CONFIGS['REF']
    g = torch.Generator().manual_seed(seed)
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=seed, bda_aug=True)]
    img = torch.randn(B, 6, 3, 256, 704, generator=g).to(dev)
    gt_depth = torch.rand(B, 6, 256, 704, generator=g) * 40 + 2
    gt_depth[torch.rand(gt_depth.shape, generator=g) > 0.03] = 0                 # ~3 % LiDAR returns
    gt_occ = torch.randint(1, 19, (B, 200, 200, 16), generator=g)
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.6] = 18                     # mostly free space
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.4] = 255                    # ~40 % not visible
    ego = torch.eye(4)
    ego[0, 3] = 0.5

    def metas(first):
        return [dict(sequence_group_idx=b, start_of_sequence=first, curr_to_prev_ego_rt=ego, index=b) for b in range(B)]
    return [img] + cam, metas, gt_occ.to(dev), gt_depth.to(dev)


def ev_ms(fn, n):
    ts = []
    for i in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], ts[max(0, len(ts) // 10)], ts[min(len(ts) - 1, len(ts) * 9 // 10)]


def infer(B, dtype, mfma=False):
    dev = torch.device('cuda:0')
    m = build(dtype, mfma=mfma).to(dev).eval()
    m.do_history = True
    img_inputs, metas, _, _ = inputs(B, dev)
    out = {}
    with torch.no_grad():
        m.predict_occupancy(img_inputs, metas(True))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.0:                                    # clock ramp-up under sustained load
            ids = m.predict_occupancy(img_inputs, metas(False))
        torch.cuda.synchronize()
        med, p10, p90 = ev_ms(lambda i: m.predict_occupancy(img_inputs, metas(False)), 10)
        # per-stage split
        t_img = ev_ms(lambda i: m.image_encoder(img_inputs[0]), 5)[0]
        x = m.image_encoder(img_inputs[0])
        cam = img_inputs[1:7]
        mlp = m.depth_net.get_mlp_input(*cam)
        t_dn = ev_ms(lambda i: m.depth_net(x, mlp), 5)[0]
        ctx, dep = m.depth_net(x, mlp)
        t_vt = ev_ms(lambda i: m.view_transform(cam, ctx.float(), dep.float()), 5)[0]
        bev = m.view_transform(cam, ctx.float(), dep.float())
        t_h = ev_ms(lambda i: m.history.fuse_history(bev, metas(False), img_inputs[6]), 5)[0]
        fused = m.history.fuse_history(bev, metas(False), img_inputs[6])
        if mfma:
            from fb_bev_amd.mfma_conv3d import to_ndhwc
            bb, neck, head = m._mfma_stacks()[:3]
            t_enc = ev_ms(lambda i: neck(bb(to_ndhwc(fused))), 5)[0]
            feats = neck(bb(to_ndhwc(fused)))
            t_head = ev_ms(lambda i: head(feats).softmax(1).argmax(1), 5)[0]
        else:
            t_enc = ev_ms(lambda i: m.bev_encoder(fused), 5)[0]
            feats = m.bev_encoder(fused)
            t_head = ev_ms(lambda i: m.occupancy_head(feats)['output_voxels'][0].softmax(1).argmax(1), 5)[0]
        if B == 1:
            t_host = ev_ms(lambda i: m.simple_test(None, metas(False), img_inputs)[0]['pred_occupancy'], 5)[0]
            out['ms_simple_test_incl_d2h'] = round(t_host, 3)
    out.update(scope='S4 full forward', B=B, conv_dtype=dtype, mfma_conv3d=mfma, pred=list(ids.shape), ms_frame=round(med, 3),
               ms_p10_p90=[round(p10, 3), round(p90, 3)], samples_per_s=round(1e3 * B / med, 2),
               ms_image_encoder=round(t_img, 3), ms_depth_net=round(t_dn, 3), ms_view_transform=round(t_vt, 3),
               ms_history_fusion=round(t_h, 3), ms_voxel_encoder=round(t_enc, 3), ms_head_argmax=round(t_head, 3),
               peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    print(json.dumps(out))


def train(B, dtype, mfma=False):
    dev = torch.device('cuda:0')
    m = build(dtype, mfma_train=mfma).to(dev).train()
    img_inputs, metas, gt_occ, gt_depth = inputs(B, dev)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-2)                  # cfg :360-362

    def step(i, first=False):
        opt.zero_grad(set_to_none=True)
        losses = m(return_loss=True, img_inputs=img_inputs, img_metas=metas(first), gt_occupancy=gt_occ, gt_depth=gt_depth)
        total = m.parse_losses(losses)
        total.backward()
        shard.finish_allreduce(shard.allreduce_gradients(params, async_op=True))
        torch.nn.utils.clip_grad_norm_(params, max_norm=5, norm_type=2)
        opt.step()
        return total, losses

    total, losses = step(0, first=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        total, losses = step(1)
    torch.cuda.synchronize()
    med, p10, p90 = ev_ms(step, 8)
    # host-sync audit of one whole step
    torch.cuda.set_sync_debug_mode('error')
    try:
        step(2)
        sync_free = True
        why = None
    except RuntimeError as e:
        sync_free, why = False, str(e).splitlines()[0][:200]
    finally:
        torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()
    print(json.dumps(dict(scope='S5 training step', B=B, conv_dtype=dtype, mfma_conv3d_train=mfma, ms_step=round(med, 3),
                          ms_p10_p90=[round(p10, 3), round(p90, 3)], samples_per_s=round(1e3 * B / med, 2),
                          loss=round(float(total), 4), losses={k: round(float(v), 4) for k, v in losses.items()},
                          step_without_host_sync=sync_free, first_sync=why, n_params=sum(p.numel() for p in params),
                          peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))))


if __name__ == '__main__':
    if 'tune' in sys.argv:          # let the vendor library search its convolution algorithms during the warm-up
        torch.backends.cudnn.benchmark = True
        sys.argv.remove('tune')
    mode = sys.argv[1] if len(sys.argv) > 1 else 'infer'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dtype = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    if mode == 'infer':
        infer(B, dtype, mfma={'mfma': True, 'mfma_bf16': 'bf16', 'mfma_bf16_tiled': 'bf16_tiled'}.get(sys.argv[4] if len(sys.argv) > 4 else '', False))
    else:
        train(B, dtype, mfma=len(sys.argv) > 4 and sys.argv[4] == 'mfma')
