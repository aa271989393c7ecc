"""One-kernel BEV self-attention (fbbev_msda_self_fused) against torch grid_sample at shapes from one workgroup per CU to two and
more (the shape class where a kernel bug that depends on co-resident workgroups shows): prints max error, the bad queries and, for
the first of them, the per-point contributions.  GPU only:  python tools/dbg_selfatt.py"""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fb_bev_amd import _capi
dev = torch.device('cuda:0')
M, Dh, P = 8, 10, 4
Em = M * Dh
for B, bh, bw in ((1, 160, 160), (1, 160, 160), (1, 200, 200), (2, 200, 200), (2, 200, 200), (1, 100, 400), (1, 400, 100), (4, 200, 200)):
    Q = bh * bw
    g = torch.Generator().manual_seed(Q + B)
    query = torch.randn(B, Q, Em, generator=g).to(dev)
    w_v, b_v = (torch.randn(Em, Em, generator=g) * 0.2).to(dev), (torch.randn(Em, generator=g) * 0.1).to(dev)
    w_so, b_so = (torch.randn(M * P * 2, Em, generator=g) * 0.15).to(dev), (torch.randn(M * P * 2, generator=g) * 2.0).to(dev)
    w_aw, b_aw = (torch.randn(M * P, Em, generator=g) * 0.2).to(dev), torch.randn(M * P, generator=g).to(dev)
    xs, ys = (torch.arange(bw) + 0.5) / bw, (torch.arange(bh) + 0.5) / bh
    ref = torch.stack([xs[None].expand(bh, bw), ys[:, None].expand(bh, bw)], -1).reshape(1, Q, 1, 2).expand(B, Q, 1, 2).contiguous().to(dev)
    value = F.linear(query, w_v, b_v).view(B, Q, M, Dh)
    so = F.linear(query, w_so, b_so).view(B, Q, M, P, 2)
    aw = F.linear(query, w_aw, b_aw).view(B, Q, M, P).softmax(-1)
    loc = ref[:, :, None, :, :] + so / torch.tensor([bw, bh], dtype=torch.float32, device=dev)       # (B, Q, M, P, 2)
    v = value.permute(0, 2, 3, 1).reshape(B * M, Dh, bh, bw)
    grid = (2 * loc - 1).permute(0, 2, 1, 3, 4).reshape(B * M, Q, P, 2)
    samp = F.grid_sample(v, grid, mode='bilinear', padding_mode='zeros', align_corners=False)      # (B*M, Dh, Q, P)
    exp = (samp * aw.permute(0, 2, 1, 3).reshape(B * M, 1, Q, P)).sum(-1).view(B, M, Dh, Q).permute(0, 3, 1, 2).reshape(B, Q, Em)
    planes = value.permute(0, 2, 1, 3).contiguous()
    out = torch.full((B, Q, Em), float('nan'), device=dev)
    f_so, f_aw = _capi.rows_linear_x3_fragments(w_so), _capi.rows_linear_x3_fragments(w_aw)
    _capi.msda_self_fused(planes, ref, query, None, f_so, b_so, f_aw, b_aw, P, bw, (bh, bw), out)
    d = (out - exp).abs()
    bad = (d.amax(-1) > 1e-3).nonzero()
    print(B, bh, bw, 'max err', d.max().item(), 'bad queries', bad.shape[0], [(int(b), int(q) // bw, int(q) % bw) for b, q in bad[:12].tolist()],
          'bad heads', sorted(set((d.view(B, Q, M, Dh).amax(-1) > 1e-3).nonzero()[:, 2].tolist()))[:8], flush=True)

    if bad.shape[0]:
        dh = d.view(B, Q, M, Dh).amax(-1)
        b0, q0, m0 = (dh > 1e-3).nonzero()[0].tolist()
        print('  first bad (b, q, head)', b0, q0, m0, 'y, x', q0 // bw, q0 % bw)
        print('  out', [round(float(v), 4) for v in out.view(B, Q, M, Dh)[b0, q0, m0]])
        print('  exp', [round(float(v), 4) for v in exp.view(B, Q, M, Dh)[b0, q0, m0]])
        l = loc[b0, q0, m0]                                # (P, 2) normalised
        print('  pix (x, y)', [(round(float(x) * bw - 0.5, 3), round(float(y) * bh - 0.5, 3)) for x, y in l], 'w', [round(float(a), 4) for a in aw[b0, q0, m0]])
        contrib = samp.view(B, M, Dh, Q, P)[b0, m0, :, q0, :] * aw[b0, q0, m0][None]          # (Dh, P)
        diff = (out.view(B, Q, M, Dh)[b0, q0, m0] - exp.view(B, Q, M, Dh)[b0, q0, m0])
        for pp in range(P):
            print('  point', pp, 'contribution', [round(float(v), 4) for v in contrib[:, pp]])
        print('  diff', [round(float(v), 4) for v in diff])
