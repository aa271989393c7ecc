"""Why does the final dense pooling kernel take ~205 us inside the S3 step and ~150 us alone (BASELINE configs[2], B = 4)?  HIP-event
time of the pooled_volume call inside the step: as is / with its gather sources (index tensors, depth, feature rows) READ once right
before it (they were written ~1 ms earlier, 0.7 GB of intermediates ago) / with the refined BEV read once.  python tools/dbg_pool_in_step.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import synthetic as S
dev = torch.device('cuda:0')
d = S.fb_path_step('BL2', 4, 4, dev, train=False)
m, cam, ctx, depth, mlvl = d['model'], d['cam'], d['ctx'], d['depth'], d['mlvl']
fp = m.forward_projection
real = fp.pooled_volume
mode = {'touch': 0}
ev = []


def patched(parts, addend=None):
    idx, dp, feat, tile_ws = parts
    if mode['touch'] & 1:
        for t in (idx.ranks_depth, idx.ranks_feat, idx.interval_starts, idx.interval_lengths, idx.interval_rank):
            t.sum()
    if mode['touch'] & 2:
        dp.sum(); feat.sum()
    if mode['touch'] & 4 and addend is not None:
        addend.sum()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = real(parts, addend=addend)
    b.record()
    ev.append((a, b))
    return r


fp.pooled_volume = patched
with torch.no_grad():
    for touch in (0, 1, 2, 3, 4, 7, 0):
        mode['touch'] = touch
        for _ in range(5):
            m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        ev.clear()
        for _ in range(20):
            m(cam, ctx, depth, mlvl_feats=mlvl)
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        print(json.dumps({'touched_before_pool': {0: 'nothing', 1: 'index tensors', 2: 'depth + feat', 3: 'index + depth + feat', 4: 'refined BEV',
                                                  7: 'everything'}[touch], 'pool_call_us_median': round(ts[len(ts) // 2], 1)}), flush=True)
