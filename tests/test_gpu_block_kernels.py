"""Kernel-level GPU parity of the encoder-block kernels that became the inference default in round 4: every
entry is called through the C ABI exactly as the module calls it and compared with the CPU oracle / an fp64 restatement, at the
sizes the path runs them at.  Observed errors are printed (collected into profiles/r05_gpu_tests_observed.txt by the session
script); the bars are stated in each test.

  fbbev_msda_self_fused / _ln ........ mmcv MultiScaleDeformableAttention.forward as bevformer_encoder.py:327-341 calls it
                                       (oracle: backward_projection_oracle.mmcv_msda_self_attention, pinned on
                                       tests/golden/mmcv_msda_forward_trt_twin.npz)
  fbbev_rows_ffn_x3 .................. mmcv FFN + the layer's LayerNorm (bevformer_encoder.py:250-377)
  fbbev_rows_linear_x3_ln ............ output_proj + residual + LayerNorm tail of both attention blocks
  fbbev_tokens_from_nchw_levels ...... bevformer.py:95-117 (flatten / permute / cams_embeds / cat)
  fbbev_pool_zmean[_split] ........... fbocc.py:359 (bev_feat.mean(-1)) without the volume
  fbbev_da_cross_attn_fused_ln ....... DA_SpatialCrossAttention + output_proj + residual + norm (opt-in route)
"""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _say(msg):
    print('[observed] ' + msg)


# ------------------------------------------------------------------ BEV self-attention in one kernel
def _self_attn_case(B, bh, bw, seed):
    M, Dh, P = 8, 10, 4
    E, Q = M * Dh, bh * bw
    g = torch.Generator().manual_seed(seed)
    pre = 'a.'
    Pm = {}
    for name, o, sw, sb in (('value_proj', E, 0.2, 0.1), ('sampling_offsets', M * P * 2, 0.15, 2.0), ('attention_weights', M * P, 0.2, 1.0),
                            ('output_proj', E, 0.2, 0.1)):
        Pm[pre + name + '.weight'] = torch.randn(o, E, generator=g) * sw
        Pm[pre + name + '.bias'] = torch.randn(o, generator=g) * sb
    Pm['n.weight'] = torch.rand(E, generator=g) + 0.5
    Pm['n.bias'] = torch.randn(E, generator=g) * 0.1
    query = torch.randn(B, Q, E, generator=g)
    pos = torch.randn(Q, E, generator=g) * 0.5
    return Pm, pre, query, pos, (M, Dh, P, E, Q)


@pytest.mark.parametrize('B,bh,bw', [(1, 100, 100), (2, 200, 200), (2, 37, 21)])
def test_one_kernel_bev_self_attention_vs_oracle_composite(dev, B, bh, bw):
    """fbbev_msda_self_fused and fbbev_msda_self_fused_ln called as MultiScaleDeformableAttention.forward calls them on the default
    inference route (fb_bev_amd/backward_projection.py: head planes from fbbev_rows_linear_x3_planes, the (Q, E) positional table as
    addend, bev_w, one level = the BEV grid, 4 points), at Q = 100 x 100 (the shipped grid), 200 x 200 (BASELINE configs[2]) and a
    grid that is no multiple of the 8 x 8 patch, against the oracle's self-attention composite (value_proj, offsets, softmax,
    bilinear sampling, output_proj, + identity; then the layer's LayerNorm for `_ln`): <= 1e-4 of the output scale."""
    from fb_bev_amd import _capi
    from oracle import backward_projection_oracle as BO
    Pm, pre, query, pos, (M, Dh, P, E, Q) = _self_attn_case(B, bh, bw, seed=bh * bw + B)
    ref2d = BO.reference_points_2d(bh, bw, B)                                             # (B, Q, 1, 2), bevformer_encoder.py:78-89
    exp = BO.mmcv_msda_self_attention(Pm, pre, query, pos[None].expand(B, Q, E), ref2d, torch.tensor([[bh, bw]]), torch.tensor([0]),
                                      num_heads=M, num_levels=1, num_points=P)            # output_proj(attention) + identity
    exp_ln = F.layer_norm(exp, (E,), Pm['n.weight'], Pm['n.bias'], 1e-5)
    g = lambda t: t.to(dev).contiguous()  # noqa: E731
    assert _capi.msda_self_fused_supported(B, Q, M, Dh, 1, Q, P, bw)
    frag = {n: _capi.rows_linear_x3_fragments(g(Pm[pre + n + '.weight'])) for n in ('value_proj', 'sampling_offsets', 'attention_weights',
                                                                                     'output_proj')}
    q_g, pos_g, ref_g = g(query), g(pos), g(ref2d)
    planes = _capi.rows_linear_x3_planes(q_g.view(B * Q, E), frag['value_proj'], g(Pm[pre + 'value_proj.bias']), Q, M, Dh)
    value = F.linear(query, Pm[pre + 'value_proj.weight'], Pm[pre + 'value_proj.bias']).view(B, Q, M, Dh).permute(0, 2, 1, 3)
    perr = (planes.cpu() - value).abs().max().item() / value.abs().max().item()
    _say(f'self-attention value planes [{B}x{bh}x{bw}]: max|err| / scale = {perr:.3e}')
    assert perr <= 2e-5
    scale = max(exp.abs().max().item(), 1.0)
    for tag, q_in, add in (('table', q_g, pos_g), ('no addend', g(query + pos[None]), None)):
        out = torch.full((B, Q, E), float('nan'), device=dev)
        _capi.msda_self_fused(planes, ref_g, q_in, add, frag['sampling_offsets'], g(Pm[pre + 'sampling_offsets.bias']),
                              frag['attention_weights'], g(Pm[pre + 'attention_weights.bias']), P, bw, (bh, bw), out)
        assert not torch.isnan(out).any(), tag
        # the oracle's composite continues with output_proj + identity: taken in fp64 on the kernel's attention output
        got = (F.linear(out.cpu().double(), Pm[pre + 'output_proj.weight'].double(), Pm[pre + 'output_proj.bias'].double())
               + query.double()).float()
        err = (got - exp).abs().max().item()
        _say(f'fbbev_msda_self_fused [{B}x{bh}x{bw} / {tag}]: max|err| vs oracle composite = {err:.3e} (output scale {scale:.2f})')
        assert err <= 1e-4 * scale, (tag, err)
        y = torch.full((B, Q, E), float('nan'), device=dev)
        _capi.msda_self_fused(planes, ref_g, q_in, add, frag['sampling_offsets'], g(Pm[pre + 'sampling_offsets.bias']),
                              frag['attention_weights'], g(Pm[pre + 'attention_weights.bias']), P, bw, (bh, bw), y,
                              out_proj=(frag['output_proj'], g(Pm[pre + 'output_proj.bias']), q_g, g(Pm['n.weight']), g(Pm['n.bias']), 1e-5))
        assert not torch.isnan(y).any(), tag
        lscale = max(exp_ln.abs().max().item(), 1.0)
        err = (y.cpu() - exp_ln).abs().max().item()
        _say(f'fbbev_msda_self_fused_ln [{B}x{bh}x{bw} / {tag}]: max|err| vs LayerNorm(oracle composite) = {err:.3e} (scale {lscale:.2f})')
        assert err <= 1e-4 * lscale, (tag, err)


# ------------------------------------------------------------------ FFN pair in one kernel / GEMM + LayerNorm epilogue
def _ffn_check(dev, rows, I, H, O, ln, with_res, tag=''):
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(rows + H + O)
    x = torch.randn(rows, I, generator=g)
    w1, b1 = torch.randn(H, I, generator=g) / I ** 0.5, torch.randn(H, generator=g) * 0.3
    w2, b2 = torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g) * 0.3
    res = torch.randn(rows, O, generator=g) if with_res else None
    lw, lb = (torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2) if ln else (None, None)
    d = lambda t: None if t is None else t.double()  # noqa: E731
    y = F.linear(torch.relu(F.linear(d(x), d(w1), d(b1))), d(w2), d(b2))                  # mmcv FFN in fp64
    if with_res:
        y = y + d(res)
    ref = F.layer_norm(y, (O,), d(lw), d(lb), 1e-5) if ln else y
    gg = lambda t: None if t is None else t.to(dev).contiguous()  # noqa: E731
    f1, f2 = _capi.rows_linear_x3_fragments(gg(w1)), _capi.rows_linear_x3_fragments(gg(w2))
    out = _capi.rows_ffn_x3(gg(x), f1, gg(b1), f2, gg(b2), H, O, residual=gg(res), ln_weight=gg(lw), ln_bias=gg(lb), eps=1e-5)
    assert not torch.isnan(out).any()
    scale = max(1.0, ref.abs().max().item())
    err = (out.cpu().double() - ref).abs().max().item()
    _say(f'fbbev_rows_ffn_x3 [{rows} rows, {I}->{H}->{O}, ln={ln}, residual={with_res}{tag}]: max|err| vs fp64 = {err:.3e} (scale {scale:.2f})')
    assert err <= 2e-5 * scale, err
    return err


FFN_CASES = [(160000, 80, 320, 80, True, True), (160000, 80, 320, 80, False, True), (40000, 80, 320, 80, True, False), (1037, 64, 128, 48, True, True)]


@pytest.mark.parametrize('rows,I,H,O,ln,with_res', FFN_CASES)
def test_ffn_one_kernel_vs_fp64(dev, rows, I, H, O, ln, with_res):
    """fbbev_rows_ffn_x3 as FFN._one_kernel calls it (fragments of both weights, bias rows, optional residual and LayerNorm), the
    default instantiation (32 hidden units per chunk), at the encoder's row count for BASELINE configs[2] B = 4 (160 000 rows)
    and a row count that does not fill the last tile: <= 2e-5 of the output scale against the fp64 composition."""
    _ffn_check(dev, rows, I, H, O, ln, with_res)


def test_ffn_one_kernel_64_unit_chunks_vs_fp64(dev):
    """The HC = 64 instantiation (FBBEV_FFN_HC=64; the knob is read once per process, hence a child process): the same bars."""
    code = ('import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_block_kernels as T\n'
            'for c in T.FFN_CASES: T._ffn_check(torch.device("cuda:0"), *c, tag=", HC=64")\n') % (ROOT, os.path.join(ROOT, 'tests'))
    env = dict(os.environ, FBBEV_FFN_HC='64')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count('[observed]') == len(FFN_CASES)


@pytest.mark.parametrize('rows,I,O,with_res', [(160000, 80, 80, True), (160000, 320, 80, True), (40000, 80, 80, False), (1037, 80, 64, True)])
def test_linear_layernorm_epilogue_vs_fp64(dev, rows, I, O, with_res):
    """fbbev_rows_linear_x3_ln (LayerNorm(x W^T + b [+ residual]) in the GEMM's store epilogue: the output_proj tail of both
    attention blocks and the 320 -> 80 tail of the two-kernel FFN) at 160 000 rows: <= 2e-5 of the output scale against fp64."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(rows + I)
    x = torch.randn(rows, I, generator=g)
    w, b = torch.randn(O, I, generator=g) / I ** 0.5, torch.randn(O, generator=g) * 0.3
    res = torch.randn(rows, O, generator=g) if with_res else None
    lw, lb = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2
    y = F.linear(x.double(), w.double(), b.double())
    ref = F.layer_norm(y + res.double() if with_res else y, (O,), lw.double(), lb.double(), 1e-5)
    gg = lambda t: None if t is None else t.to(dev).contiguous()  # noqa: E731
    out = _capi.rows_linear_x3_ln(gg(x), _capi.rows_linear_x3_fragments(gg(w)), gg(b), O, gg(res), gg(lw), gg(lb), 1e-5)
    assert not torch.isnan(out).any()
    scale = max(1.0, ref.abs().max().item())
    err = (out.cpu().double() - ref).abs().max().item()
    _say(f'fbbev_rows_linear_x3_ln [{rows} rows, {I}->{O}, residual={with_res}]: max|err| vs fp64 = {err:.3e} (scale {scale:.2f})')
    assert err <= 2e-5 * scale, err


@pytest.mark.parametrize('rows,E,H,with_res', [(160000, 80, 320, True), (40000, 80, 320, False), (1037, 64, 128, True)])
def test_attention_tail_plus_ffn_one_kernel_vs_fp64(dev, rows, E, H, with_res):
    """fbbev_rows_tail_ffn_x3 (round 5: output_proj + residual + norm of the cross-attention block AND the FFN block + norm,
    bevformer_encoder.py:250-377, one kernel) as DA_SpatialCrossAttention.forward calls it, at the encoder's row count for BASELINE
    configs[2] B = 4: <= 2e-5 of the output scale against the fp64 composition, and <= 1e-5 of it against the two kernels it
    replaces (fbbev_rows_linear_x3_ln -> fbbev_rows_ffn_x3)."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(rows + H + E)
    x = torch.randn(rows, E, generator=g)
    w0, b0 = torch.randn(E, E, generator=g) / E ** 0.5, torch.randn(E, generator=g) * 0.3
    res0 = torch.randn(rows, E, generator=g) if with_res else None
    l0w, l0b = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g) * 0.2
    w1, b1 = torch.randn(H, E, generator=g) / E ** 0.5, torch.randn(H, generator=g) * 0.3
    w2, b2 = torch.randn(E, H, generator=g) / H ** 0.5, torch.randn(E, generator=g) * 0.3
    l1w, l1b = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g) * 0.2
    d = lambda t: None if t is None else t.double()  # noqa: E731
    y0 = F.linear(d(x), d(w0), d(b0))
    y1 = F.layer_norm(y0 + d(res0) if with_res else y0, (E,), d(l0w), d(l0b), 1e-5)
    ref = F.layer_norm(y1 + F.linear(torch.relu(F.linear(y1, d(w1), d(b1))), d(w2), d(b2)), (E,), d(l1w), d(l1b), 1e-5)
    gg = lambda t: None if t is None else t.to(dev).contiguous()  # noqa: E731
    f0, f1, f2 = (_capi.rows_linear_x3_fragments(gg(w)) for w in (w0, w1, w2))
    xg, rg = gg(x), gg(res0)
    assert _capi.rows_tail_ffn_x3_supported(xg, rg, E, H)
    out = _capi.rows_tail_ffn_x3(xg, f0, gg(b0), rg, gg(l0w), gg(l0b), 1e-5, f1, gg(b1), f2, gg(b2), H, gg(l1w), gg(l1b), 1e-5)
    assert not torch.isnan(out).any()
    t1 = _capi.rows_linear_x3_ln(xg, f0, gg(b0), E, rg, gg(l0w), gg(l0b), 1e-5)
    two = _capi.rows_ffn_x3(t1, f1, gg(b1), f2, gg(b2), H, E, residual=t1, ln_weight=gg(l1w), ln_bias=gg(l1b), eps=1e-5)
    scale = max(1.0, ref.abs().max().item())
    err = (out.cpu().double() - ref).abs().max().item()
    dlt = (out - two).abs().max().item()
    _say(f'fbbev_rows_tail_ffn_x3 [{rows} rows, E={E}, H={H}, residual={with_res}]: max|err| vs fp64 = {err:.3e}, vs the two kernels it '
         f'replaces = {dlt:.3e} (scale {scale:.2f})')
    assert err <= 2e-5 * scale and dlt <= 1e-5 * scale, (err, dlt)
    assert not _capi.rows_tail_ffn_x3_supported(xg[:, :E - 8], None, E - 8, H)          # embed no multiple of 16
    assert not _capi.rows_tail_ffn_x3_supported(xg.view(-1)[2:2 + (rows - 1) * E].view(rows - 1, E), None, E, H)   # misaligned view


def test_encoder_layer_with_and_without_the_fused_tail_ffn_route(dev):
    """BackwardProjection through the module with the cross-attention tail + FFN as one kernel (default) and as the two round-4 kernels
    (FBBEV_FUSE_TAIL_FFN=0): the same output within the split-operand arithmetic (1e-5 of scale), and the default route really runs
    fbbev_rows_tail_ffn_x3."""
    from fb_bev_amd import _capi, backward_projection as BP
    from test_gpu_backward_projection import _setup
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=2, num_levels=4, bev=40, shapes=[(16, 44), (32, 88), (8, 22), (4, 11)])
    args = ([f.to(dev) for f in feats], None)
    kw = dict(lss_bev=lss.to(dev), cam_params=[t.to(dev) for t in cam], pred_img_depth=depth.to(dev))
    calls = []
    real = _capi.rows_tail_ffn_x3
    _capi.rows_tail_ffn_x3 = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            one = m(*args, **kw)
            assert calls, 'the default inference route did not take fbbev_rows_tail_ffn_x3'
            BP.FUSE_TAIL_FFN = False
            n = len(calls)
            two = m(*args, **kw)
            assert len(calls) == n
    finally:
        BP.FUSE_TAIL_FFN = True
        _capi.rows_tail_ffn_x3 = real
    scale = max(1.0, two.abs().max().item())
    dlt = (one - two).abs().max().item()
    _say(f'BackwardProjection, tail + FFN as one kernel vs two: max|diff| = {dlt:.3e} (scale {scale:.2f})')
    assert dlt <= 1e-5 * scale


def test_side_stream_prefetch_of_the_backward_projection_changes_no_bit(dev):
    """FBViewTransform (inference) starts the Z-mean-independent part of the backward projection -- camera-token rows, their value
    planes, the BEV -> image point sampling -- on a side stream under the forward projection's ranking chain (round 5,
    BackwardProjection.prefetch).  Same kernels, same inputs: the output is bit-identical to the single-stream order
    (FBBEV_BP_PREFETCH=0), every one of 20 back-to-back calls with changing inputs (a missed stream dependency would show as a
    stale or half-written buffer), and the prefetch really is consumed."""
    from fb_bev_amd import backward_projection as BP, configs, synthetic as S
    from fb_bev_amd.fb_view_transform import FBViewTransform
    pc = S.CONFIGS['BL2']
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample, num_levels=4)
    torch.manual_seed(0)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection'])
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if 'sampling_offsets.weight' in n_ or 'attention_weights.weight' in n_:
                p_.normal_(0, 0.05)
    m = m.to(dev).eval()
    B = 2
    H, W = pc.feat_hw
    shapes = [(H, W), (2 * H, 2 * W), (H // 2, W // 2), (H // 4, W // 4)]
    used = []
    real = BP.BackwardProjection.prefetch

    def spy(self, *a, **k):
        r = real(self, *a, **k)
        used.append(r is not None and bool(r.planes) and r.rows is not None and r.sampling is not None)
        return r
    BP.BackwardProjection.prefetch = spy
    min_q, BP.PREFETCH_MIN_QUERIES = BP.PREFETCH_MIN_QUERIES, 0          # (the module only takes the route for >= 120 000 queries)
    try:
        with torch.no_grad():
            for it in range(20):
                cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=it, bda_aug=True)]
                depth, ctx = (t.to(dev) for t in S.depth_and_context(pc, B, seed=it))
                g = torch.Generator().manual_seed(100 + it)
                mlvl = [ctx] + [torch.randn(B, pc.n_cams, pc.channels, h, w, generator=g).to(dev) for h, w in shapes[1:]]
                BP.PREFETCH = True
                a = m(cam, ctx, depth, mlvl_feats=mlvl)
                BP.PREFETCH = False
                b = m(cam, ctx, depth, mlvl_feats=mlvl)
                assert torch.equal(a, b), (it, (a - b).abs().max().item())
    finally:
        BP.PREFETCH = True
        BP.PREFETCH_MIN_QUERIES = min_q
        BP.BackwardProjection.prefetch = real
    assert used.count(True) == 20, used
    _say('BackwardProjection.prefetch on a side stream: 20 / 20 calls bit-identical to the single-stream order')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_one_kernel_da_cross_attention_on_16bit_head_planes(dev, dt):
    """Round 5: the camera-token storage option (DA_SpatialCrossAttention.value_dtype) on the one-kernel route: value_proj writes bf16 /
    fp16 head planes (fbbev_rows_linear_x3_planes_e == the fp32 planes rounded once), fbbev_da_cross_attn_fused_e samples them with
    fp32 products and sums: EXACTLY the fp32 kernel's slots on the widened planes (Q = 100 x 100, the configs[2] pyramid), and within
    the storage rounding of the oracle composite on fp32 tokens (the reference keeps fp32: an option, not the default)."""
    from da_cases import da_case
    from fb_bev_amd import _capi
    args, exp, ex = da_case(12, E=80, M=8, P=8, extras=True, B=1, Q=10000, shapes=((32, 88), (16, 44), (8, 22), (4, 11)), DC=59)
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    BN, S_, M, Dh = value.shape
    Pm, pre = ex['Pm'], 'a.deformable_attention.'
    g = lambda t: t.to(dev).contiguous()  # noqa: E731
    frag = {n: _capi.rows_linear_x3_fragments(g(Pm[pre + n + '.weight'])) for n in ('value_proj', 'sampling_offsets', 'attention_weights')}
    x = g(ex['key'].permute(2, 0, 1, 3).reshape(BN * S_, M * Dh))
    p32 = _capi.rows_linear_x3_planes(x, frag['value_proj'], g(Pm[pre + 'value_proj.bias']), S_, M, Dh)
    p16 = _capi.rows_linear_x3_planes(x, frag['value_proj'], g(Pm[pre + 'value_proj.bias']), S_, M, Dh, dtype=dt)
    assert p16.dtype == dt and torch.equal(p16, p32.to(dt))
    common = (g(ss), g(ls), g(pred), g(ref_cam), g(mask), g(qdepth), g(ex['query']), g(ex['qpos'].reshape(-1, M * Dh)),
              frag['sampling_offsets'], g(Pm[pre + 'sampling_offsets.bias']), frag['attention_weights'], g(Pm[pre + 'attention_weights.bias']),
              8, d0, dstep, 100, 11)
    s16 = _capi.da_cross_attn_fused(p16, *common, torch.full(exp.shape, float('nan'), device=dev))
    swide = _capi.da_cross_attn_fused(p16.float().contiguous(), *common, torch.full(exp.shape, float('nan'), device=dev))
    assert not torch.isnan(s16).any() and torch.equal(s16, swide)
    err = (s16.cpu() - exp).abs().max().item()
    scale = max(exp.abs().max().item(), 1.0)
    _say(f'fbbev_da_cross_attn_fused_e [{str(dt)[6:]} head planes]: == the fp32 kernel on the widened planes bit for bit; max|err| vs the oracle '
         f'composite on fp32 tokens = {err:.3e} (scale {scale:.2f}: storage rounding)')
    assert err <= (1.5e-2 if dt == torch.bfloat16 else 2e-3) * scale


# ------------------------------------------------------------------ camera-token pyramid in one launch
@pytest.mark.parametrize('images,C,shapes', [(24, 80, ((16, 44), (32, 88), (8, 22), (4, 11))), (6, 80, ((16, 44),)), (5, 33, ((5, 9), (8, 4), (1, 3), (2, 2)))])
def test_token_pyramid_in_one_launch_bit_exact(dev, images, C, shapes):
    """fbbev_tokens_from_nchw_levels as BEVFormer.forward calls it == flatten(3).permute + cams_embeds + cat (bevformer.py:95-117),
    bit for bit: the BASELINE configs[2] pyramid for B = 4 (24 images), the shipped single level, odd sizes; with / without the
    camera embedding rows."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(images + C)
    levels = [torch.randn(images, C, h * w, generator=g).to(dev) for h, w in shapes]
    S_ = sum(h * w for h, w in shapes)
    for ncam_bias in (0, 6 if images % 6 == 0 else images):
        bias = torch.randn(ncam_bias, C, generator=g).to(dev) if ncam_bias else None
        out = torch.full((images, S_, C), float('nan'), device=dev)
        _capi.tokens_from_nchw_levels(levels, out, bias=bias)
        exp = torch.cat([t.permute(0, 2, 1) for t in levels], 1)
        if bias is not None:
            exp = exp + bias[torch.arange(images, device=dev) % ncam_bias][:, None, :]
        assert torch.equal(out, exp), (images, C, shapes, ncam_bias)
    _say(f'fbbev_tokens_from_nchw_levels [{images} images, C={C}, {len(shapes)} levels]: bit-exact vs permute + cat')


# ------------------------------------------------------------------ Z-mean without the volume, one workgroup per Z plane
@pytest.mark.parametrize('name,B', [('REF', 1), ('REF', 4), ('BL2', 1), ('SMALL', 2)])
def test_pool_zmean_split_equals_single_pass_and_oracle_volume_mean(dev, name, B):
    """fbbev_pool_zmean_split (every Z plane of a pixel tile its own workgroup + ordered reduce; FBViewTransform's default for
    grids with few tiles, fbocc.py:359) against fbbev_pool_zmean: with one plane per group (z_groups = Z, the module's choice) the
    SAME BITS -- both add the per-plane fmaf chains of a pillar to a running sum in Z order --, with several planes per group another
    association of that sum (<= 1e-6 of the volume scale apart); bit-identical run to run; and both within 1e-5 of the volume scale
    of the mean over Z of the ORACLE's pooled volume (oracle/fbbev_oracle.c, the reference kernel's fmaf chain per voxel)."""
    from fb_bev_amd import _capi, synthetic as S
    from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
    from oracle import oracle as O
    cfg = S.CONFIGS[name]
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    cam = S.camera_rig(cfg, B, seed=3, bda_aug=True)
    depth, ctx = S.depth_and_context(cfg, B, seed=3)
    vt = LSSViewTransformerFunction3D(cfg.grid_config, cfg.input_size, cfg.downsample).to(dev)
    cam_g = [t.to(dev) for t in cam]
    parts = vt.pooling_inputs(cam_g, ctx.to(dev), depth.to(dev))
    idx, d_g, f_g, tile_ws = parts
    Z, Y, X = vt.grid_zyx
    C = cfg.channels
    tv = vt._wo_tile
    one = torch.full((B, C, Y, X), float('nan'), device=dev)
    _capi.pool_zmean(d_g, f_g, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts, idx.interval_lengths,
                     B, C, Z, Y, X, one, tile_ws, tv, vt.pool_flags)
    assert not torch.isnan(one).any()
    coor = vt.get_lidar_coor(*cam_g).cpu()                                               # contract pinned at the ranking input
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(coor)
    vol = O.bev_pool_v2(depth, ctx.permute(0, 1, 3, 4, 2).contiguous(), rd, rf, rb, ovt.bev_feat_shape(B, C), st, ln)   # (B,C,Z,Y,X)
    exp = vol.double().mean(2)
    scale = max(vol.abs().max().item(), 1.0)
    err1 = (one.cpu().double() - exp).abs().max().item()
    assert err1 <= 1e-5 * scale, err1
    for zg in sorted({Z, max(2, Z // 2)}):
        outs = []
        for _ in range(2):
            split = torch.full((B, C, Y, X), float('nan'), device=dev)
            partial = torch.full((zg * split.numel(),), float('nan'), device=dev)
            _capi.pool_zmean(d_g, f_g, idx.ranks_depth, idx.ranks_feat, idx.interval_rank, idx.interval_starts, idx.interval_lengths,
                             B, C, Z, Y, X, split, tile_ws, tv, vt.pool_flags, z_groups=zg, partial=partial)
            assert not torch.isnan(split).any()
            outs.append(split)
        assert torch.equal(outs[0], outs[1])                                             # deterministic
        d = (split - one).abs().max().item()
        err = (split.cpu().double() - exp).abs().max().item()
        _say(f'fbbev_pool_zmean_split [{name} B={B}, z_groups {zg} of Z={Z}]: max|split - single pass| = {d:.3e}, max|err| vs mean_z(oracle '
             f'volume) = {err:.3e} (single pass {err1:.3e}; volume scale {scale:.2f})')
        assert err <= 1e-5 * scale and (d == 0.0 if zg == Z else d <= 1e-6 * scale), (zg, d, err)      # one plane per group: the same bits
    # the module's own route (split form for grids with <= 1 024 tiles) stays inside the same bar
    assert (vt.pooled_zmean(parts).cpu().double() - exp).abs().max().item() <= 1e-5 * scale


def test_pool_zmean_split_more_planes_than_the_entry_takes(dev):
    """a grid with more than 64 Z planes must not raise on the default Z-mean route (fbbev_pool_zmean_split refuses
    z_groups > 64): the module falls back to the single pass."""
    from fb_bev_amd.view_transformer import LSSViewTransformerFunction3D
    from fb_bev_amd import synthetic as S
    cfg = S.CONFIGS['SMALL']
    gc = dict(cfg.grid_config)
    gc['z'] = [-1.0, 8.0, 0.125]
    vt = LSSViewTransformerFunction3D(gc, cfg.input_size, cfg.downsample).to(dev)
    assert vt.grid_zyx[0] == 72
    cam = S.camera_rig(cfg, 1, seed=0, bda_aug=False)
    depth, ctx = S.depth_and_context(cfg, 1, seed=0)
    parts = vt.pooling_inputs([t.to(dev) for t in cam], ctx.to(dev), depth.to(dev))
    zm = vt.pooled_zmean(parts)
    vol = vt.pooled_volume(parts)                                                         # (B,C,Y,X,Z) view
    assert torch.allclose(zm, vol.mean(-1), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ cross-attention block tail inside the sampler (opt-in route)
@pytest.mark.parametrize('case', ['shipped', 'bl3_pyramid', 'partial_patches'])
def test_one_kernel_da_cross_attention_with_block_tail_vs_oracle_composite(dev, case):
    """fbbev_da_cross_attn_fused_ln (FBBEV_FUSE_ATTN_TAIL_DA=1): LayerNorm(output_proj(slots) + residual) in the sampler's
    workgroups, called as DA_SpatialCrossAttention calls it, against the oracle's DA_SpatialCrossAttention.forward INCLUDING its
    output_proj + residual (spatial_cross_attention_depth.py:136-223) followed by the layer's LayerNorm: <= 1e-4 of scale."""
    from da_cases import da_case
    from fb_bev_amd import _capi
    from oracle import backward_projection_oracle as BO
    kw, bev_w = dict(shipped=(dict(B=2, Q=10000, shapes=((16, 44),), DC=80), 100),
                     bl3_pyramid=(dict(B=1, Q=10000, shapes=((32, 88), (16, 44), (8, 22), (4, 11)), DC=59), 100),
                     partial_patches=(dict(B=2, Q=37 * 21, shapes=((16, 44), (8, 22)), DC=30), 21))[case]
    args, slots_exp, ex = da_case(13, E=80, M=8, P=8, extras=True, **kw)
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    BN, S_, M, Dh = value.shape
    E = M * Dh
    Pm, pre = ex['Pm'], 'a.deformable_attention.'
    gen = torch.Generator().manual_seed(5)
    w_o, b_o = torch.randn(E, E, generator=gen) * 0.2, torch.randn(E, generator=gen) * 0.1
    lnw, lnb = torch.rand(E, generator=gen) + 0.5, torch.randn(E, generator=gen) * 0.1
    exp = F.layer_norm(F.linear(slots_exp, w_o, b_o) + ex['query'], (E,), lnw, lnb, 1e-5)   # :222-223 + norms.1
    g = lambda t: t.to(dev).contiguous()  # noqa: E731
    frag = {n: _capi.rows_linear_x3_fragments(g(Pm[pre + n + '.weight'])) for n in ('value_proj', 'sampling_offsets', 'attention_weights')}
    x = g(ex['key'].permute(2, 0, 1, 3).reshape(BN * S_, E))
    planes = _capi.rows_linear_x3_planes(x, frag['value_proj'], g(Pm[pre + 'value_proj.bias']), S_, M, Dh)
    out = torch.full(exp.shape, float('nan'), device=dev)
    _capi.da_cross_attn_fused(planes, g(ss), g(ls), g(pred), g(ref_cam), g(mask), g(qdepth), g(ex['query']), g(ex['qpos'].reshape(-1, E)),
                              frag['sampling_offsets'], g(Pm[pre + 'sampling_offsets.bias']), frag['attention_weights'],
                              g(Pm[pre + 'attention_weights.bias']), 8, d0, dstep, bev_w, min(w for _, w in kw['shapes']), out,
                              out_proj=(_capi.rows_linear_x3_fragments(g(w_o)), g(b_o), g(ex['query']), g(lnw), g(lnb), 1e-5))
    assert not torch.isnan(out).any()
    scale = max(exp.abs().max().item(), 1.0)
    err = (out.cpu() - exp).abs().max().item()
    _say(f'fbbev_da_cross_attn_fused_ln [{case}]: max|err| vs LayerNorm(oracle DA block) = {err:.3e} (scale {scale:.2f})')
    assert err <= 1e-4 * scale, err
