"""Is the one-kernel DA sampler slowed by cold inputs inside the S3 step (BASELINE configs[2], B = 4)?  HIP-event time of the
fbbev_da_cross_attn_fused call with some of its inputs read once (fbbev_touch) right before it.  python tools/dbg_da_in_step.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import _capi, synthetic as S
dev = torch.device('cuda:0')
d = S.fb_path_step('BL2', 4, 4, dev, train=False)
m, cam, ctx, depth, mlvl = d['model'], d['cam'], d['ctx'], d['depth'], d['mlvl']
real = _capi.da_cross_attn_fused
mode = {'touch': 0}
ev = []


def patched(planes, ss, ls, pred_depth, ref_cam, mask, qdepth, query, *a, **k):
    t = mode['touch']
    if t & 1:
        _capi.touch(planes)
    if t & 2:
        _capi.touch(ref_cam, mask.view(torch.uint8), qdepth)
    if t & 4:
        _capi.touch(pred_depth, query)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(planes, ss, ls, pred_depth, ref_cam, mask, qdepth, query, *a, **k)
    e1.record()
    ev.append((e0, e1))
    return r


_capi.da_cross_attn_fused = patched
with torch.no_grad():
    for touch in (0, 1, 2, 4, 7, 0):
        mode['touch'] = touch
        for _ in range(5):
            m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        ev.clear()
        for _ in range(20):
            m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        print(json.dumps({'touched_before_da': {0: 'nothing', 1: 'camera-token head planes (115 MB)', 2: 'reference points / masks / query depths (52 MB)',
                                                4: 'depth distribution + query rows', 7: 'everything'}[touch],
                          'da_call_us_median': round(ts[len(ts) // 2], 1)}), flush=True)
