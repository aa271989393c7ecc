"""GPU parity of the backward projection against the CPU oracle
(oracle/backward_projection_oracle.py, pinned on fixtures from the real reference Python).
fp32 tolerance 1e-4 on O(1) features (the path is floating point; the MSDA op is pinned on the reference
tree's twin of mmcv's bilinear functions, tests/test_oracle_msda_ref.py -- see DESIGN.md section 4)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
# full-size BackwardProjection test: the bars (set from the observed values, see the test)
# observed on an MI355X (profiles/r04_gpu_tests_observed.txt): 0 of 40 000 queries beyond 1e-3, max|err| 7.5e-5 among the rest, no
# element beyond 1e-4, median 4.4e-6 at an output scale of 4.8.  Bars = 2x observed; the query allowance stays above zero because
# an in-image mask bit that flips between the CPU and GPU op orders is a property of the rig, not of the kernels.
BP_FULL_BAD_QUERIES = 4
BP_FULL_MAX_ERR = 1e-4          # round 6: north_star's own bar (observed 7.5e-5), not 2x observed
BP_FULL_FRAC_1E4 = 0.0
sys.path.insert(0, os.path.dirname(__file__))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_fused_da_kernel_vs_oracle_composite(dev):
    from test_emu_kernels import _da_case
    from fb_bev_amd import _capi
    for seed, kw in ((0, {}), (1, dict(B=1, Q=33, shapes=((4, 6),), P=4, M=2, E=8)),
                     (2, dict(B=2, Q=2000, E=80, M=8, shapes=((16, 44),), P=8, DC=80)),          # shipped shapes
                     (3, dict(B=1, Q=500, E=80, M=8, shapes=((32, 88), (16, 44), (8, 22), (4, 11)), P=8, DC=59))):  # BL3
        args, exp = _da_case(seed, **kw)
        value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        g = lambda t: t.to(dev).contiguous()  # noqa: E731
        slots = torch.full(exp.shape, float('nan'), device=dev)
        _capi.da_cross_attn_fwd(g(value), g(ss), g(ls), g(pred), g(ref_cam), g(mask), g(qdepth), g(offsets), g(attn),
                                d0, dstep, slots)
        assert not torch.isnan(slots).any()
        assert torch.allclose(slots.cpu(), exp, atol=1e-4, rtol=1e-4), (seed, (slots.cpu() - exp).abs().max())


def _interleave_rows(v, HS):
    """(B*N, S, M, Dh) head-major tokens -> the chunk-major padded rows of the module's value projection: each head padded to HS
    floats, a token's floats stored (HS/4, M, 4) (backward_projection._pad_interleave_rows applied to the weight rows)."""
    BN, S_, M, Dh = v.shape
    vp = torch.zeros(BN, S_, M, HS)
    vp[..., :Dh] = v
    return vp.view(BN, S_, M, HS // 4, 4).transpose(-3, -2).contiguous().view(BN * S_, M * HS)


@pytest.mark.parametrize('case', ['shipped', 'bl3_pyramid'])
def test_default_pipelined_da_kernel_vs_oracle_composite(dev, case):
    """VERDICT r3 (weak 1-i): the kernel the module runs by default in inference -- fbbev_da_cross_attn_fwd_zt ->
    k_da_cross_attn_fwd_pipe -- called EXACTLY as DA_SpatialCrossAttention._slots_fused calls it (rows in `da_value_buffer` with
    the zero token behind them, chunk-major head-padded tokens, head-minor offsets: head_minor = 5; with / without
    FBBEV_DA_ATTN_LOGITS; linear unit order and the 2-D patch mapping `bev_w`), at the shipped shape (Q = 100 x 100, one 16x44
    level, 80 depth bins) and at the BASELINE configs[2] pyramid (4 levels, Q = 100 x 100), <= 1e-4 against the oracle's
    composite (spatial_cross_attention_depth.py:136-223, 513-595).  A poisoned zero token must change the result (padded
    corners read it) and stay finite."""
    from da_cases import da_case
    from fb_bev_amd import _capi
    kw = dict(shipped=dict(B=2, Q=10000, E=80, M=8, shapes=((16, 44),), P=8, DC=80),
              bl3_pyramid=dict(B=1, Q=10000, E=80, M=8, shapes=((32, 88), (16, 44), (8, 22), (4, 11)), P=8, DC=59))[case]
    args, exp = da_case(11, **kw)
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    BN, S_, M, Dh = value.shape
    HS = 12
    g = lambda t: t.to(dev).contiguous()  # noqa: E731
    buf, rows = _capi.da_value_buffer(BN * S_, M * HS, dev)
    rows.copy_(_interleave_rows(value, HS))
    off_hm = g(offsets.permute(0, 1, 3, 4, 2, 5))                       # (B,Q,L,P,M,2)
    B, Q = attn.shape[:2]
    L, P = attn.shape[3:]
    logits = (attn.flatten(-2).log() + torch.randn(B, Q, M, 1, generator=torch.Generator().manual_seed(3)) * 3
              ).view(attn.shape)                                        # softmax(log p + c) == p
    common = (g(ss), g(ls), g(pred), g(ref_cam), g(mask), g(qdepth), off_hm)
    assert _capi.da_fuses_softmax(B, 6, S_, M, Dh, L, Q, P, 4, 5, HS)    # the pipelined kernel takes this shape
    results = {}
    for tag, a, hm, bw in (('linear', attn, 5, 0), ('patch', attn, 5, 100), ('patch+logits', logits, 5 | _capi.DA_ATTN_LOGITS, 100),
                           ('logits', logits, 5 | _capi.DA_ATTN_LOGITS, 0)):
        slots = torch.full(exp.shape, float('nan'), device=dev)
        _capi.da_cross_attn_fwd(rows.view(BN, S_, M, HS), *common, g(a), d0, dstep, slots, head_minor=hm, head_dim=Dh,
                                zero_token=True, bev_w=bw)
        assert not torch.isnan(slots).any(), tag
        err = (slots.cpu() - exp).abs().max().item()
        print(f'pipelined DA kernel [{case} / {tag}]: max|err| vs oracle composite = {err:.3e}')
        assert torch.allclose(slots.cpu(), exp, atol=1e-4, rtol=1e-4), (tag, err)
        results[tag] = slots
    assert torch.equal(results['linear'], results['patch'])             # only the lane -> unit map changes
    assert torch.equal(results['logits'], results['patch+logits'])
    buf[BN * S_].fill_(3.0)                                             # poisoned zero token
    slots = torch.empty(exp.shape, device=dev)
    _capi.da_cross_attn_fwd(rows.view(BN, S_, M, HS), *common, g(attn), d0, dstep, slots, head_minor=5, head_dim=Dh,
                            zero_token=True, bev_w=100)
    assert torch.isfinite(slots).all() and not torch.equal(slots, results['patch'])


@pytest.mark.parametrize('case', ['shipped', 'bl3_pyramid', 'partial_patches'])
def test_one_kernel_da_cross_attention_vs_oracle_composite(dev, case):
    """fbbev_da_cross_attn_fused (round 4, the inference default): query rows -> slots in one kernel -- in-kernel sampling_offsets /
    attention_weights projections (split-operand bf16 MFMA), softmax in LDS, head-plane camera tokens written by
    fbbev_rows_linear_x3_planes -- called as DA_SpatialCrossAttention._slots_one_kernel calls it, at the shipped shape
    (Q = 100 x 100, one 16x44 level, 80 bins), the BASELINE configs[2] pyramid and a grid that is no multiple of the 8 x 8 patch:
    <= 1e-4 of the output scale against the oracle's composite (spatial_cross_attention_depth.py:136-223, 513-595)."""
    from da_cases import da_case
    from fb_bev_amd import _capi
    kw, bev_w = dict(shipped=(dict(B=2, Q=10000, shapes=((16, 44),), DC=80), 100),
                     bl3_pyramid=(dict(B=1, Q=10000, shapes=((32, 88), (16, 44), (8, 22), (4, 11)), DC=59), 100),
                     partial_patches=(dict(B=2, Q=37 * 21, shapes=((16, 44), (8, 22)), DC=30), 21))[case]
    args, exp, ex = da_case(12, E=80, M=8, P=8, extras=True, **kw)
    value, ss, ls, pred, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
    BN, S_, M, Dh = value.shape
    Pm, pre = ex['Pm'], 'a.deformable_attention.'
    g = lambda t: t.to(dev).contiguous()  # noqa: E731
    assert _capi.da_cross_attn_fused_supported(ex['query'].shape[0], 6, S_, M, Dh, len(kw['shapes']), kw['Q'], 8, 4, bev_w)
    frag = {n: _capi.rows_linear_x3_fragments(g(Pm[pre + n + '.weight'])) for n in ('value_proj', 'sampling_offsets', 'attention_weights')}
    x = g(ex['key'].permute(2, 0, 1, 3).reshape(BN * S_, M * Dh))
    planes = _capi.rows_linear_x3_planes(x, frag['value_proj'], g(Pm[pre + 'value_proj.bias']), S_, M, Dh)
    exact = _capi.rows_to_head_planes(g(value.reshape(BN * S_, M * Dh)), S_, M, Dh)
    assert torch.equal(exact.cpu(), value.permute(0, 2, 1, 3))
    assert (planes - exact).abs().max().item() <= 2e-5 * exact.abs().max().item()         # split-operand arithmetic
    scale = max(exp.abs().max().item(), 1.0)
    for tag, q, add in (('table', ex['query'], ex['qpos'].reshape(-1, M * Dh)), ('no addend', ex['query'] + ex['qpos'], None)):
        slots = torch.full(exp.shape, float('nan'), device=dev)
        _capi.da_cross_attn_fused(planes, g(ss), g(ls), g(pred), g(ref_cam), g(mask), g(qdepth), g(q), None if add is None else g(add),
                                  frag['sampling_offsets'], g(Pm[pre + 'sampling_offsets.bias']), frag['attention_weights'],
                                  g(Pm[pre + 'attention_weights.bias']), 8, d0, dstep, bev_w, min(w for _, w in kw['shapes']), slots)
        assert not torch.isnan(slots).any(), tag
        err = (slots.cpu() - exp).abs().max().item()
        print(f'one-kernel DA [{case} / {tag}]: max|err| vs oracle composite = {err:.3e} (output scale {scale:.2f})')
        assert err <= 1e-4 * scale, (tag, err)


def _setup(dev, B=2, num_levels=1, bev=20, seed=0, shapes=None):
    from fb_bev_amd import backward_projection as BP, configs, synthetic as S
    gcb = {'x': [-40, 40, 80.0 / bev], 'y': [-40, 40, 80.0 / bev], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(num_levels=num_levels, bev_h=bev, bev_w=bev, grid_config_bevformer=gcb)
    torch.manual_seed(seed)
    m = BP.build(cfg['backward_projection'])
    # non-trivial weights so offsets / attention depend on the query (the reference init zeroes them)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if 'sampling_offsets.weight' in name or 'attention_weights.weight' in name:
                p.normal_(0, 0.05)
    m = m.to(dev).eval()
    pcfg = S.CONFIGS['REF']
    cam = S.camera_rig(pcfg, B, seed=seed, bda_aug=True)
    g = torch.Generator().manual_seed(seed + 5)
    C, DC = 80, 80
    shapes = list(shapes) if shapes is not None else [(16, 44), (8, 22), (4, 11), (2, 6)][:num_levels]
    feats = [torch.randn(B, 6, C, h, w, generator=g) for h, w in shapes]
    depth = (torch.randn(B, 6, DC, 16, 44, generator=g) * 3).softmax(2)
    lss = torch.randn(B, C, bev, bev, generator=g)
    return m, cfg, cam, feats, depth, lss, gcb


def _oracle_out(m, cfg, cam, feats, depth, lss, gcb, bev, num_levels):
    from oracle import backward_projection_oracle as BO, oracle as O
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return BO.backward_projection(P, feats, lss, cam, depth, bev, bev, gcb, (256, 704), cfg['depth_bound'],
                                  inverse=O.inv3x3_closed_form)


def test_last_layer_writes_the_refined_bev_as_planes_itself(dev, monkeypatch):
    """Inference: the last encoder layer's tail + FFN kernel stores the refined BEV as (B, C, Y, X) (fbbev_rows_tail_ffn_x3_planes) instead
    of rows + the transposing pass of backward_projection.py:129 -- the SAME BITS as that route (FBBEV_BP_OUT_PLANES=0), and the
    route is actually taken (no transposing launch left behind the encoder)."""
    from fb_bev_amd import backward_projection as BP, _capi
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, num_levels=4, bev=20)
    args = dict(lss_bev=lss.to(dev), cam_params=[t.to(dev) for t in cam], pred_img_depth=depth.to(dev))
    calls = []
    real = _capi.transpose_last2
    monkeypatch.setattr(_capi, 'transpose_last2', lambda x: (calls.append(1), real(x))[1])
    with torch.no_grad():
        monkeypatch.setattr(BP, 'OUT_PLANES', False)
        rows_route = m([f.to(dev) for f in feats], None, **args)
        n_rows = len(calls)
        monkeypatch.setattr(BP, 'OUT_PLANES', True)
        planes_route = m([f.to(dev) for f in feats], None, **args)
    assert n_rows == 1 and len(calls) == 1                       # the planes route made no transposing call
    assert planes_route.shape == rows_route.shape and planes_route.is_contiguous()
    assert torch.equal(planes_route, rows_route)


@pytest.mark.parametrize('num_levels', [1, 4])
def test_backward_projection_module_vs_oracle(dev, num_levels):
    bev = 20
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, num_levels=num_levels, bev=bev)
    with torch.no_grad():                                        # -> fused kernel path
        out = m([f.to(dev) for f in feats], None, lss_bev=lss.to(dev), cam_params=[t.to(dev) for t in cam],
                pred_img_depth=depth.to(dev))
    exp = _oracle_out(m, cfg, cam, feats, depth, lss, gcb, bev, num_levels)
    assert out.shape == exp.shape == (2, 80, bev, bev)
    err = (out.cpu() - exp).abs()
    # point_sampling masks are compared through the output: a borderline projected point may flip its mask bit
    # between CPU and GPU fp32 op orders, which changes that single query -> allow a handful of such queries
    bad = (err > 1e-3).any(dim=1).sum().item()
    assert bad <= 3, bad
    assert err.median().item() < 1e-5


def test_backward_projection_module_vs_oracle_at_baseline_config2_full_size(dev):
    """VERDICT r2 (untested sizes): BASELINE configs[2] through the MODULE at its full size -- bev 200x200 (Q = 40 000), the
    4-level pyramid 16x44 / 32x88 / 8x22 / 4x11 (level 0 = the depth net's level, spatial_cross_attention_depth.py:586),
    B = 1 -- against oracle/backward_projection_oracle.py (backward_projection.py:84-133, bevformer_encoder.py:250-377).
    Bar: <= 1e-4 absolute on every element (north_star's bar; output scale 4.8; observed 7.5e-5) except at most 4 queries whose borderline
    in-image mask bit flips between the CPU and GPU fp32 op orders (observed: none of 40 000); the statistics are printed."""
    bev = 200
    shapes = [(16, 44), (32, 88), (8, 22), (4, 11)]
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=1, num_levels=4, bev=bev, shapes=shapes)
    with torch.no_grad():
        out = m([f.to(dev) for f in feats], None, lss_bev=lss.to(dev), cam_params=[t.to(dev) for t in cam],
                pred_img_depth=depth.to(dev))
    exp = _oracle_out(m, cfg, cam, feats, depth, lss, gcb, bev, 4)
    assert out.shape == exp.shape == (1, 80, bev, bev)
    err = (out.cpu() - exp).abs()
    bad = (err > 1e-3).any(dim=1).sum().item()
    ok = ~(err > 1e-3).any(dim=1, keepdim=True).expand_as(err)
    frac4 = (err[ok] > 1e-4).float().mean().item()
    print(f'BackwardProjection full size vs oracle: queries beyond 1e-3 (mask-bit flips) = {bad} of {bev * bev}; among the rest '
          f'max|err| = {err[ok].max().item():.3e}, fraction of elements beyond 1e-4 = {frac4:.2e}, median = {err.median().item():.2e}, '
          f'output scale = {exp.abs().max().item():.3f}')
    # the bars are 2x what this test printed on an MI355X (profiles/r04_gpu_tests_observed.txt), not round numbers
    assert bad <= BP_FULL_BAD_QUERIES, bad
    assert err[ok].max().item() <= BP_FULL_MAX_ERR and err.median().item() < 1e-5
    assert frac4 <= BP_FULL_FRAC_1E4, frac4


@pytest.mark.parametrize('train_fused', [True, False])
def test_training_paths_equal_inference_and_backprop(dev, train_fused):
    """train_fused=True: FusedDACrossAttention (fbbev_da_cross_attn_fwd + _bwd, no host sync);
    False: the composite rebatch + MSDA-op path.  Both against the fused inference output and the oracle's autograd."""
    from oracle import backward_projection_oracle as BO, oracle as O
    from fb_bev_amd.backward_projection import DA_SpatialCrossAttention
    bev = 12
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=1, num_levels=1, bev=bev, seed=3)
    cam_g = [t.to(dev) for t in cam]
    with torch.no_grad():
        fused = m([f.to(dev) for f in feats], None, lss_bev=lss.to(dev), cam_params=cam_g, pred_img_depth=depth.to(dev))
    for mod in m.modules():
        if isinstance(mod, DA_SpatialCrossAttention):
            mod.fused = train_fused
    f_g = [f.to(dev).requires_grad_() for f in feats]
    d_g = depth.to(dev).requires_grad_()
    l_g = lss.to(dev).requires_grad_()
    comp = m(f_g, None, lss_bev=l_g, cam_params=cam_g, pred_img_depth=d_g)       # grad enabled -> training path
    assert torch.allclose(comp, fused, atol=1e-4, rtol=1e-4)
    w = torch.randn(comp.shape, generator=torch.Generator().manual_seed(9))
    (comp * w.to(dev)).sum().backward()
    # oracle autograd (grid_sample formulation) on CPU, double precision
    P = {k: v.detach().cpu().double().requires_grad_() for k, v in m.state_dict().items()}
    f_c = [f.double().requires_grad_() for f in feats]
    d_c = depth.double().requires_grad_()
    l_c = lss.double().requires_grad_()
    out_c = BO.backward_projection(P, f_c, l_c, cam, d_c,   # fp32 cameras: same in-image masks as the GPU
                                   bev, bev, gcb, (256, 704),
                                   cfg['depth_bound'], inverse=O.inv3x3_closed_form)
    (out_c * w.double()).sum().backward()
    def close(a, b, tol):
        # fp32 GPU vs fp64 oracle: elementwise within tol of the tensor scale for >= 98 % of the entries and
        # never wildly off.  Isolated kinks are legitimate: a ReLU pre-activation or a bilinear cell boundary
        # within fp32 rounding of zero flips for ONE BEV query (1/144 of the entries here; diagnosed with
        # tools/diag_bp_grad.py: every other entry agrees to ~1e-3).
        scale = b.abs().max().item() + 1e-12
        err = (a.double().cpu() - b).abs() / scale
        return (err > tol).double().mean().item() <= 0.02 and err.max().item() <= 0.05
    assert close(l_g.grad, l_c.grad, 2e-3)
    assert close(f_g[0].grad, f_c[0].grad, 2e-3)
    assert close(d_g.grad, d_c.grad, 5e-3)
    for name, p in m.named_parameters():
        if p.grad is not None and P[name].grad is not None and P[name].grad.abs().max() > 0:
            assert close(p.grad, P[name].grad, 5e-3), name


def test_only_offset_and_attention_heads_trainable_still_get_gradients(dev):
    """with grad enabled, value_proj frozen and no input requiring grad (fine-tuning only the
    sampling_offsets / attention_weights heads), DA_SpatialCrossAttention._slots_fused used to take the inference-only zero-token
    branch -- slots without an autograd node, the two heads silently got no gradient.  Their gradients must exist and equal the
    composite (reference-shaped) path's."""
    from fb_bev_amd.backward_projection import DA_SpatialCrossAttention
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=1, num_levels=1, bev=12, seed=6)
    cam_g = [t.to(dev) for t in cam]
    for name, p in m.named_parameters():
        p.requires_grad_('sampling_offsets' in name or 'attention_weights' in name)
    das = [x for x in m.modules() if isinstance(x, DA_SpatialCrossAttention)]
    assert das
    w = torch.randn(1, 80, 12, 12, generator=torch.Generator().manual_seed(9)).to(dev)
    grads = {}
    for fused in (True, False):
        for x in das:
            x.fused = fused
        m.zero_grad(set_to_none=True)
        out = m([f.to(dev) for f in feats], None, lss_bev=lss.to(dev), cam_params=cam_g, pred_img_depth=depth.to(dev))
        assert out.requires_grad
        (out * w).sum().backward()
        grads[fused] = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}
    n_da = 0
    for n, gf in grads[True].items():
        gc = grads[False][n]
        assert gf is not None and torch.isfinite(gf).all()
        scale = gc.abs().max().item()
        if scale > 0:
            assert (gf - gc).abs().max().item() <= 5e-3 * scale + 1e-6, (n, (gf - gc).abs().max().item(), scale)
        n_da += int(gf.abs().max().item() > 0)
    assert n_da >= 2, 'no sampling_offsets / attention_weights parameter received a non-zero gradient'


def test_row_functions_survive_autocast(dev):
    """_RowsLinear / _LayerNormRows under a user-level torch.autocast (the reference trains under mmcv's fp16
    hook): the custom Functions cast to fp32 and switch autocast off inside forward and backward -- before, backward mixed a
    bf16 grad_output with fp32 operands and raised."""
    from fb_bev_amd import rows_linear as RL
    from fb_bev_amd.backward_projection import LayerNorm
    torch.manual_seed(0)
    lin = RL.Linear(80, 128).to(dev)
    ln = LayerNorm(80).to(dev)
    x = torch.randn(RL.MIN_ROWS + 64, 80, device=dev, requires_grad=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = lin(ln(x))
        assert y.dtype == torch.float32
        loss = (y.float() ** 2).mean()
    loss.backward()
    g_ac = [x.grad.clone(), lin.weight.grad.clone(), ln.weight.grad.clone()]
    x.grad = None; lin.zero_grad(); ln.zero_grad()
    (lin(ln(x)) ** 2).mean().backward()
    for a, b in zip(g_ac, [x.grad, lin.weight.grad, ln.weight.grad]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('dt,tol', [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)])
def test_16bit_camera_tokens_vs_fp32_tokens(dev, dt, tol):
    """DA_SpatialCrossAttention.value_dtype: camera tokens rounded once to 16 bits, fp32 accumulate -- on the one-kernel route
    (16-bit head planes, fbbev_da_cross_attn_fused_e; round 5) and, with FBBEV_DA_16BIT_PLANES=0, on the round-3 kernels
    (fbbev_da_cross_attn_fwd_e).  The stated error is against the fp32-token path on the same inputs: max |diff| relative to the
    output scale (the emulator tests pin the kernels themselves bit for bit on the rounded tokens); the two 16-bit routes read the
    same rounded tokens and differ by the fp32 association only."""
    from fb_bev_amd import backward_projection as BP
    from fb_bev_amd.backward_projection import DA_SpatialCrossAttention
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=2, num_levels=2, bev=20, seed=4)
    cam_g = [t.to(dev) for t in cam]
    args = ([f.to(dev) for f in feats], None)
    kw = dict(lss_bev=lss.to(dev), cam_params=cam_g, pred_img_depth=depth.to(dev))
    with torch.no_grad():
        ref = m(*args, **kw)
        mods = [x for x in m.modules() if isinstance(x, DA_SpatialCrossAttention)]
        assert mods
        for x in mods:
            x.value_dtype = dt
        seen, orig = [], BP._capi.da_cross_attn_fused
        BP._capi.da_cross_attn_fused = lambda planes, *a, **k: (seen.append(planes.dtype), orig(planes, *a, **k))[1]
        keep = BP.PLANES_16BIT
        try:
            got = m(*args, **kw)
            assert seen and all(d == dt for d in seen), seen      # the one-kernel sampler ran, on 16-bit head planes
            del seen[:]
            BP.PLANES_16BIT = False
            got_r3 = m(*args, **kw)
            assert not seen
        finally:
            BP.PLANES_16BIT, BP._capi.da_cross_attn_fused = keep, orig
        for x in mods:
            x.value_dtype = None
        again = m(*args, **kw)
    assert torch.equal(again, ref)                       # the option leaves no state behind
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    assert 0 < err < tol, err
    assert 0 < (got_r3 - ref).abs().max().item() / scale < tol
    assert (got - got_r3).abs().max().item() / scale < 1e-4


def test_graphed_replay_equals_eager_for_new_inputs(dev):
    """fb_bev_amd.graphed.Graphed: the whole forward + backward projection captured once and replayed with OTHER camera
    rigs / features / depth than the captured example -- equal to the eager call bit for bit (the index tensors are
    rebuilt on the device inside the graph)."""
    from fb_bev_amd import configs, synthetic as S
    from fb_bev_amd.fb_view_transform import FBViewTransform
    from fb_bev_amd.graphed import Graphed
    pc = S.CONFIGS['REF']
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample)
    torch.manual_seed(0)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(dev).eval()

    def inputs(seed):
        cam = [t.to(dev) for t in S.camera_rig(pc, 1, seed=seed, bda_aug=True)]
        depth, ctx = (t.to(dev) for t in S.depth_and_context(pc, 1, seed=seed))
        return cam, ctx, depth
    g = Graphed(m, *inputs(0))
    for seed in (0, 3, 4):
        cam, ctx, depth = inputs(seed)
        with torch.no_grad():
            eager = m(cam, ctx, depth)
        assert torch.equal(g(cam, ctx, depth), eager), seed
    with pytest.raises(ValueError):
        g(cam, ctx[:, :3], depth)


@pytest.mark.parametrize('lds_planes', [False, True])
def test_fused_backward_kernels_within_a_bound_of_fp64_autograd(dev, lds_planes):
    """A BOUND, not a statistical pass (the module-level test above tolerates 2 % kinked entries): the four gradients of
    fbbev_da_cross_attn_bwd / _bwd_ws, pushed back to the leaves, against the double-precision autograd of the oracle's
    loop-for-loop restatement of the reference's training path -- EVERY entry within 1e-4 relative + 5e-5 of the tensor
    scale.  The comparison is at the kernel boundary (no ReLU / LayerNorm around it), so no kink can flip."""
    from fb_bev_amd import _capi
    from da_cases import da_case
    for seed, kw in ((5, dict(B=1, Q=29, E=16, M=4)),
                     (6, dict(B=2, Q=17, E=40, M=4, shapes=((4, 6), (2, 3)))),
                     (7, dict(B=2, Q=333, E=80, M=8, shapes=((16, 44),), DC=20)),       # the shipped head layout
                     (8, dict(B=1, Q=257, E=80, M=8, shapes=((16, 44), (32, 88), (8, 22), (4, 11)), DC=20)),   # configs[2] pyramid: 6 token regions
                     # VERDICT r2 (untested sizes): the same bound at Q >= 10 000 -- many query chunks per workgroup plan,
                     # long per-token accumulation chains in the fixed-point planes
                     (9, dict(B=1, Q=10000, E=80, M=8, shapes=((16, 44), (32, 88), (8, 22), (4, 11)), DC=20)),
                     (10, dict(B=1, Q=40000, E=80, M=8, shapes=((16, 44),), DC=20))):          # configs[2]'s Q on the shipped level
        args, exp, leaves = da_case(seed, grad=True, **kw)
        g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        wrt = [leaves['key'], leaves['pred']] + [leaves['Pm'][k] for k in sorted(leaves['Pm']) if 'output_proj' not in k]
        ref = torch.autograd.grad(exp, wrt, grad_outputs=g, retain_graph=True, allow_unused=True)
        value, ss, ls, pred4, ref_cam, mask, qdepth, offsets, attn, d0, dstep = args
        Dh = value.shape[-1]
        HS = (Dh + 3) // 4 * 4
        f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
        vp = torch.zeros(value.shape[:-1] + (HS,))
        vp[..., :Dh] = f32(value)
        t = lambda x: x.to(dev).contiguous()  # noqa: E731
        a = [t(vp), t(ss), t(ls), t(f32(pred4)), t(f32(ref_cam)), t(mask), t(f32(qdepth)), t(f32(offsets)), t(f32(attn)),
             t(f32(g)), d0, dstep, 0]
        gv, gd, go, ga = (torch.zeros_like(x) for x in (a[0], a[3], a[7], a[8]))
        _capi.da_cross_attn_bwd(*a, gv, gd, go, ga, head_dim=Dh, lds_planes=lds_planes,
                                level_hw=[tuple(int(x) for x in hw) for hw in ss.tolist()])
        mine = torch.autograd.grad([value, pred4, offsets, attn], wrt,
                                   grad_outputs=[gv[..., :Dh].cpu().double(), gd.cpu().double(), go.cpu().double(), ga.cpu().double()],
                                   retain_graph=True, allow_unused=True)
        assert not gv[..., Dh:].any()
        names = ['key', 'pred'] + [k for k in sorted(leaves['Pm']) if 'output_proj' not in k]
        for name, x, y in zip(names, mine, ref):
            if y is None:
                assert x is None or not x.any()
                continue
            # Every gradient but one keeps the same bound at any Q (measured at Q = 10 000 / 40 000: 2e-6 .. 5e-6 of the
            # scale for both kernels, profiles/r03_diag_da_bwd_bound.jsonl).  The exception is d/d(sampling offset): the
            # bilinear sample is only piecewise smooth in its location, so a sample whose fp32 location falls on the other
            # side of a cell boundary than its fp64 location contributes a DIFFERENT (finite) slope.  Such samples are a
            # fixed small fraction of the Q * M * L * P * cameras samples, and the leaves sum over all queries: the
            # deviation grows ~ linearly with Q (1e-6 at Q = 2 500, 5e-4 at 10 000, 2e-2 at 40 000 of the scale) -- the
            # fp32 reference kernel has the same property.  Allowance: 1e-6 * Q of the scale, for these two leaves only.
            kink = 1e-6 * kw.get('Q', 70) if 'sampling_offsets' in name and kw.get('Q', 70) >= 5000 else 0.0
            bound = 1e-4 * y.abs() + (5e-5 + kink) * max(1.0, y.abs().max().item())
            assert ((x - y).abs() <= bound).all(), (name, (x - y).abs().max().item(), y.abs().max().item(), seed, lds_planes)


def test_lds_plane_backward_equals_atomic_backward_and_is_reproducible(dev):
    """fbbev_da_cross_attn_bwd_ws (fixed-point gradient planes in LDS, partial buffer, fixed-order reduction) against
    fbbev_da_cross_attn_bwd (fp32 global atomics) at the shipped shape; the value gradient of the former is bit-identical
    run to run (integer adds commute), which the atomic kernel -- like mmcv's col2im -- is not required to be."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(11)
    B, Ncam, Q, M, Dh, HS, P, Za, DC, H0, W0 = 2, 6, 2500, 8, 10, 12, 8, 4, 80, 16, 44
    S_ = H0 * W0
    value = torch.randn(B * Ncam, S_, M, HS, generator=g)
    value[..., Dh:] = 0
    value = value.view(B * Ncam, S_, HS // 4, M, 4).contiguous().view(B * Ncam, S_, M, HS)      # any floats: layout-agnostic
    pred = torch.rand(B * Ncam, DC, H0, W0, generator=g).softmax(1)
    ref_cam = torch.rand(Ncam, B, Q, Za, 2, generator=g) * 1.2 - 0.1
    mask = torch.rand(Ncam, B, Q, Za, generator=g) < 0.15
    qdepth = torch.rand(Ncam, B, Q, Za, generator=g) * 45 + 1
    offsets = torch.randn(B, Q, 1, P, M, 2, generator=g) * 1.5                                  # head-minor (B,Q,L,P,M,2)
    attn = torch.rand(B, Q, M, 1 * P, generator=g).softmax(-1).view(B, Q, M, 1, P)
    gs = torch.randn(B, Q, M * Dh, generator=g) * 3.0
    ss = torch.tensor([[H0, W0]]); ls = torch.tensor([0])
    t = lambda x: x.to(dev).contiguous()  # noqa: E731
    args = [t(value), t(ss), t(ls), t(pred), t(ref_cam), t(mask), t(qdepth), t(offsets), t(attn), t(gs), 1.0, 0.5, 1 | 4]

    def run(lds):
        gv, gd, go, ga = (torch.zeros_like(x) for x in (args[0], args[3], args[7], args[8]))
        _capi.da_cross_attn_bwd(*args, gv, gd, go, ga, head_dim=Dh, lds_planes=lds)
        torch.cuda.synchronize()
        return gv, gd, go, ga
    assert _capi.da_cross_attn_bwd_ws_bytes(B, Ncam, S_, M, Dh, Q, HS, 1, P) > 0
    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0])                                      # reproducible value gradient
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])          # and the unit-owned ones
    for x, y in zip(a, c):
        scale = y.abs().max().item()
        assert scale > 0 and (x - y).abs().max().item() <= 2e-6 * scale + 1e-7, ((x - y).abs().max().item(), scale)
    # non-finite upstream gradient: the chunk's planes are NaN, not silently finite
    args[9] = args[9].clone(); args[9][0, 0, 0] = float('inf')
    assert torch.isnan(run(True)[0]).any()


def test_output_owned_plane_backward_at_a_pyramid(dev):
    """Round 4: the value-gradient scatter with output-owned LDS planes (k_da_bwd_hitlist + k_da_bwd_scatter_owned: one
    workgroup per (sample, camera, head, token region), hit lists, no partial planes) -- the route fbbev_da_cross_attn_bwd_ws
    takes when the launch has a workgroup per CU -- on a three-level pyramid whose first level is split into bands of rows:
    against the fp32-global-atomic kernel, bit-identical run to run, NaN for a non-finite upstream gradient."""
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(23)
    B, Ncam, Q, M, Dh, HS, P, Za, DC = 2, 6, 5000, 8, 10, 12, 8, 4, 40
    shapes = [(32, 88), (16, 44), (8, 22)]
    L = len(shapes)
    H0, W0 = shapes[0]
    S_ = sum(h * w for h, w in shapes)
    value = torch.randn(B * Ncam, S_, M, HS, generator=g)
    pred = torch.rand(B * Ncam, DC, H0, W0, generator=g).softmax(1)
    ref_cam = torch.rand(Ncam, B, Q, Za, 2, generator=g) * 1.2 - 0.1
    mask = torch.rand(Ncam, B, Q, Za, generator=g) < 0.15
    qdepth = torch.rand(Ncam, B, Q, Za, generator=g) * 25 + 1
    offsets = torch.randn(B, Q, L, P, M, 2, generator=g) * 2.0                                  # head-minor (B,Q,L,P,M,2)
    attn = torch.rand(B, Q, M, L * P, generator=g).softmax(-1).view(B, Q, M, L, P)
    gs = torch.randn(B, Q, M * Dh, generator=g) * 3.0
    ss = torch.tensor(shapes)
    ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    t = lambda x: x.to(dev).contiguous()  # noqa: E731
    args = [t(value), t(ss), t(ls), t(pred), t(ref_cam), t(mask), t(qdepth), t(offsets), t(attn), t(gs), 1.0, 0.5, 1 | 4]
    # the training forward on head planes (k_da_fwd_planes) against the row kernel: same samples, another summation order
    assert _capi.da_cross_attn_fwd_planes_supported(B, Ncam, S_, M, Dh, L, Q, P, Za)
    planes = _capi.value_rows_to_head_planes(args[0], head_dim=Dh, interleaved=True)
    assert torch.equal(planes.cpu(), value.view(B * Ncam, S_, HS // 4, M, 4).permute(0, 3, 1, 2, 4).reshape(B * Ncam, M, S_, HS)[..., :Dh])
    s_rows = torch.full((B, Q, M * Dh), float('nan'), device=dev)
    _capi.da_cross_attn_fwd(*args[:9], 1.0, 0.5, s_rows, head_minor=1 | 4, head_dim=Dh)
    for bw in (0, 100):
        s_pl = torch.full((B, Q, M * Dh), float('nan'), device=dev)
        _capi.da_cross_attn_fwd_planes(planes, *args[1:9], 1.0, 0.5, s_pl, head_minor=1 | 4, bev_w=bw, min_level_width=22)
        err = (s_pl - s_rows).abs().max().item()
        assert not torch.isnan(s_pl).any() and err <= 1e-5 * max(1.0, s_rows.abs().max().item()), (bw, err)
    # the owned route is planned: its workspace is the hit records (+ the planes), not partial planes
    records = B * Ncam * Q * 16 * 4                                   # 64-byte hit records in list order
    need = _capi.da_cross_attn_bwd_ws_bytes(B, Ncam, S_, M, Dh, Q, HS, L, P, level_hw=shapes, Za=mask.shape[3])
    assert _capi.da_cross_attn_bwd_ws_bytes(B, Ncam, S_, M, Dh, Q, HS, L, P, level_hw=shapes) >= need      # without Za: covers both routes
    planes = B * Ncam * M * S_ * Dh * 4                               # + the camera tokens as head planes for the unit gradients
    assert records + planes <= need <= records + planes + 5 * 256, (need, records, planes)

    def run(lds, bev_w=0):
        gv = torch.full_like(args[0], float('nan')) if lds else torch.zeros_like(args[0])      # the owned planes write every word
        gd, go, ga = (torch.zeros_like(x) for x in (args[3], args[7], args[8]))
        _capi.da_cross_attn_bwd(*args, gv, gd, go, ga, head_dim=Dh, lds_planes=lds, level_hw=shapes if lds else None, bev_w=bev_w)
        torch.cuda.synchronize()
        return gv, gd, go, ga
    a, b, c = run(True), run(True), run(False)
    # the unit gradients run on head planes here (k_da_bwd_unit_planes: M = 8, Dh = 10, 8 points, 4 anchors); with the BEV grid's
    # width they take 8 x 8 patches of the 50 x 100 queries instead of runs of 64: the same sums per unit in the same order
    d = run(True, bev_w=100)
    assert torch.equal(a[0], d[0]) and torch.equal(a[2], d[2]) and torch.equal(a[3], d[3])
    assert (a[1] - d[1]).abs().max().item() <= 2e-6 * a[1].abs().max().item()                # fp32 atomics into the depth planes
    assert not torch.isnan(a[0]).any() and not a[0].view(B * Ncam, S_, HS // 4, M, 4)[:, :, 2, :, 2:].any()   # padding channels stay 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for i, (x, y) in enumerate(zip(a, c)):
        scale = y.abs().max().item()
        # value gradient: the comparison kernel sums thousands of fp32 atomics per coarse token (its own rounding, observed
        # 1.0e-5 of the scale here); the planes are exact integer sums with one rounding
        bar = 2e-5 if i == 0 else 2e-6
        assert scale > 0 and (x - y).abs().max().item() <= bar * scale + 1e-7, (i, (x - y).abs().max().item(), scale)
    # the fixed-point scale is per SAMPLE -- an outlier 1e6 x the typical upstream gradient in sample 1 costs resolution
    # there (quantum 2^-30 of ITS maximum) and leaves sample 0's value gradient bit for bit what it was
    base = a[0].view(B, Ncam, S_, M * HS)
    args[9] = args[9].clone(); args[9][1, 7, 3] = 3.0e6
    out = run(True)[0].view(B, Ncam, S_, M * HS)
    ref = run(False)[0].view(B, Ncam, S_, M * HS)
    assert torch.equal(out[0], base[0])
    err1 = (out[1] - ref[1]).abs().max().item()
    print(f'[observed] owned-plane scatter with a 1e6x outlier in sample 1: sample 0 bit-identical; sample 1 max|err| vs the fp32-atomic kernel = '
          f'{err1:.3e} (its gradient scale {ref[1].abs().max().item():.3e}, quantum {3.0e6 * 2.0 ** -30:.1e})')
    assert err1 <= 4 * 3.0e6 * 2.0 ** -30 * 64 + 2e-5 * ref[1].abs().max().item()        # <= a few quanta per addend chain + the comparison kernel's rounding
    args[9][1, 7, 3] = float('inf')
    out = run(True)[0].view(B, Ncam, S_, M * HS)
    assert torch.isnan(out[1]).any() and torch.equal(out[0], base[0])                    # NaN in the poisoned sample only


@pytest.mark.parametrize('E,M,L', [(80, 8, 1), (64, 8, 2)])
def test_self_attention_fused_inference_equals_composed(dev, E, M, L):
    """MultiScaleDeformableAttention inference (fbbev_msda_fwd_fused: locations built in the kernel, padded value rows)
    against the composed path (torch location tensor + fbbev_msda_fwd)."""
    from fb_bev_amd.backward_projection import MultiScaleDeformableAttention
    torch.manual_seed(E)
    m = MultiScaleDeformableAttention(embed_dims=E, num_heads=M, num_levels=L, num_points=4, batch_first=True)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.05)
        m.attention_weights.weight.normal_(0, 0.05)
    m = m.to(dev).eval()
    shapes = [(20, 18), (10, 9)][:L]
    ss = torch.tensor(shapes, device=dev)
    ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    S_ = int((ss[:, 0] * ss[:, 1]).sum())
    B, Q = 2, 360
    q = torch.randn(B, Q, E, device=dev)
    v = torch.randn(B, S_, E, device=dev)
    ref = torch.rand(B, Q, L, 2, device=dev)
    with torch.no_grad():
        fused = m(q, value=v, reference_points=ref, spatial_shapes=ss, level_start_index=ls)
        m.fused_inference = False
        comp = m(q, value=v, reference_points=ref, spatial_shapes=ss, level_start_index=ls)
    assert torch.allclose(fused, comp, atol=2e-6, rtol=1e-6), (fused - comp).abs().max()


def test_point_sampling_kernel_vs_reference_python_fixture(dev):
    """fbbev_point_sampling against the fixture produced by the REAL bevformer_encoder.point_sampling."""
    import numpy as np
    from fb_bev_amd import backward_projection as BP, configs, synthetic as S
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'point_sampling_REF_B2_aug.npz'))
    cfg = configs.fbocc_r50()
    enc = BP.build(cfg['backward_projection']['transformer']['encoder']).to(dev)
    cam = [t.to(dev) for t in S.camera_rig(S.CONFIGS['REF'], 2, seed=0, bda_aug=True)]
    ref3d = enc.get_reference_points(100, 100, 6.4, dim='3d', bs=2, device=dev, dtype=torch.float)
    assert np.array_equal(ref3d[:2, :2].cpu().numpy(), z['ref3d_corner'])
    _, ref_cam, mask, qd = enc.point_sampling(ref3d, None, None, cam_params=cam)
    sub = slice(0, 10000, 37)
    m_g, m_r = mask[:, :, sub].cpu().numpy(), z['mask']
    assert (m_g != m_r).sum() <= 2                                    # borderline projections may flip
    assert abs(int(mask.sum()) - int(z['mask_count'])) <= 8
    vis = m_r & m_g
    assert np.allclose(ref_cam[:, :, sub].cpu().numpy()[vis], z['ref_cam'][vis], atol=5e-5)
    assert np.allclose(qd[:, :, sub].cpu().numpy()[vis], z['qdepth'][vis], atol=5e-4, rtol=1e-5)


def test_layernorm_kernel_vs_torch(dev):
    from fb_bev_amd import _capi
    g = torch.Generator().manual_seed(0)
    for rows, C in ((160000, 80), (1000, 128), (7, 4)):
        x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dev)
        r = torch.randn(rows, C, generator=g).to(dev)
        w, b = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        exp = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
        assert (_capi.layernorm(x, w, b, 1e-5) - exp).abs().max().item() < 1e-5
        exp2 = torch.nn.functional.layer_norm(x + r, (C,), w, b, 1e-5)
        assert (_capi.layernorm(x, w, b, 1e-5, residual=r) - exp2).abs().max().item() < 1e-5


def test_fused_training_step_has_no_host_sync(dev):
    """forward + backward of the backward projection with autograd enabled: FusedDACrossAttention, no nonzero()/max()."""
    m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=2, num_levels=1, bev=16, seed=1)
    cam_g = [t.to(dev) for t in cam]
    f_g = [f.to(dev).requires_grad_() for f in feats]
    d_g, l_g = depth.to(dev).requires_grad_(), lss.to(dev).requires_grad_()
    m(f_g, None, lss_bev=l_g, cam_params=cam_g, pred_img_depth=d_g).sum().backward()      # warm-up (allocations, caches)
    for t in f_g + [d_g, l_g]:
        t.grad = None
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        m(f_g, None, lss_bev=l_g, cam_params=cam_g, pred_img_depth=d_g).sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in f_g + [d_g, l_g])


@pytest.mark.parametrize('name,B', [('REF', 2), ('BL2', 1), ('REF', 1)])
def test_write_once_volume_path_equals_reference_composition(dev, name, B):
    """FBViewTransform inference: Z-mean from the index tensors (fbbev_pool_zmean) + re-add in the pooling store
    (fbbev_bev_pool_v2_dense_fwd_add) == pool -> mean(-1) -> backward projection -> refined[...,None] + volume."""
    from fb_bev_amd import configs, synthetic as S
    from fb_bev_amd.fb_view_transform import FBViewTransform
    pc = S.CONFIGS[name]
    X, Y, Z = pc.grid_xyz
    gcb = {'x': pc.grid_config['x'], 'y': pc.grid_config['y'], 'z': [-1, 5.4, 1.6]}
    cfg = configs.fbocc_r50(bev_h=Y, bev_w=X, numC_Trans=pc.channels, input_size=pc.input_size, grid_config=pc.grid_config,
                            grid_config_bevformer=gcb, depth_bound=tuple(pc.grid_config['depth']), downsample=pc.downsample)
    torch.manual_seed(0)
    m = FBViewTransform(cfg['forward_projection'], cfg['backward_projection']).to(dev).eval()
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=0, bda_aug=True)]
    depth, ctx = S.depth_and_context(pc, B, seed=0)
    depth, ctx = depth.to(dev), ctx.to(dev)
    with torch.no_grad():
        m.write_once = True
        a = m(cam, ctx, depth)
        m.write_once = False
        b = m(cam, ctx, depth)
        # the pieces: Z-mean and add epilogue against the materialised volume
        fp = m.forward_projection
        vol = fp(cam, ctx, depth)
        parts = fp.pooling_inputs(cam, ctx, depth)
        assert (fp.pooled_zmean(parts) - vol.mean(-1)).abs().max().item() < 1e-5
        addend = torch.randn(B, pc.channels, Y, X, device=dev)
        assert torch.equal(fp.pooled_volume(parts, addend=addend), vol + addend[..., None])
    assert a.shape == b.shape == (B, pc.channels, Y, X, Z)
    assert (a - b).abs().max().item() < 1e-4
    # round 6: the Z-mean handed over as query rows + bev_embedding (fbbev_pool_zmean_rows: no transposing pass in front of the encoder)
    # and the refined BEV written as planes by the last layer's kernel (no transposing pass behind it): the SAME BITS as with both
    # passes -- where the single-pass Z-mean is what the module runs (many tiles); with Z groups the rows form declines (None)
    from fb_bev_amd import fb_view_transform as FV, backward_projection as BPm
    with torch.no_grad():
        m.write_once = True
        rows = fp.pooled_zmean_rows(parts, m.backward_projection.query_row_bias(pc.channels, fp.grid_zyx))
        if rows is not None:
            emb = m.backward_projection.bev_embedding.weight
            assert torch.equal(rows, fp.pooled_zmean(parts).flatten(2).transpose(1, 2) + emb[None])
        old = (FV.ZMEAN_ROWS, BPm.OUT_PLANES)
        try:
            FV.ZMEAN_ROWS, BPm.OUT_PLANES = False, False
            c = m(cam, ctx, depth)
        finally:
            FV.ZMEAN_ROWS, BPm.OUT_PLANES = old
    assert torch.equal(a, c)


@pytest.mark.gpu
def test_rows_linear_split_k_backward_on_the_gpu():
    """rows_linear at the configs[2] row count (160 000 x 80 -> 128): the GPU route is taken (custom backward node) and its
    weight / bias / input gradients equal autograd's plain GEMMs to fp32 rounding."""
    from fb_bev_amd import rows_linear as RL
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 40000, 80, generator=g).to(dev).requires_grad_()
    lin = RL.Linear(80, 128).to(dev)
    gy = torch.randn(4, 40000, 128, generator=g).to(dev)
    y = lin(x)
    assert 'RowsLinear' in type(y.grad_fn.next_functions[0][0]).__name__     # y is a view of the custom node's 2-D result
    y = torch.relu_(y)                       # the FFN's in-place ReLU must be legal on the result
    y.backward(gy)
    got = [x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = lin.weight.grad = lin.bias.grad = None
    y2 = torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))
    y2.backward(gy)
    assert torch.equal(y, y2)
    for a, r in zip(got, [x.grad, lin.weight.grad, lin.bias.grad]):
        assert (a - r).abs().max() <= 2e-5 * r.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize('rows,C', [(160000, 80), (1000, 128), (37, 64)])
def test_layernorm_training_route_equals_torch_autograd(rows, C):
    """backward_projection.LayerNorm with autograd on: fbbev_layernorm + fbbev_layernorm_bwd (custom node) against
    nn.LayerNorm's own forward / backward -- output, input gradient, weight / bias gradients."""
    from fb_bev_amd.backward_projection import LayerNorm
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(rows + C)
    ln = LayerNorm(C).to(dev)
    ref = torch.nn.LayerNorm(C).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g)); ln.bias.copy_(torch.randn(C, generator=g))
    ref.load_state_dict(ln.state_dict())
    x = (torch.randn(4, rows // 4 if rows % 4 == 0 else rows, C, generator=g) * 2 + 0.5).to(dev)
    res = torch.randn(x.shape, generator=g).to(dev)
    gy = torch.randn(x.shape, generator=g).to(dev)
    for r in (None, res):
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya = ln(xa, residual=r)
        assert 'LayerNormRows' in type(ya.grad_fn).__name__
        yb = ref(xb if r is None else xb + r)
        ya.backward(gy); yb.backward(gy)
        assert torch.allclose(ya, yb, atol=3e-6, rtol=2e-5)
        assert torch.allclose(xa.grad, xb.grad, atol=5e-6, rtol=5e-5)
        for a, b in ((ln.weight.grad, ref.weight.grad), (ln.bias.grad, ref.bias.grad)):
            assert (a - b).abs().max() <= 3e-5 * b.abs().max().clamp_min(1.0)
        ln.zero_grad(); ref.zero_grad()


@pytest.mark.gpu
def test_msda_backward_band_binned_route_on_the_gpu():
    """MultiScaleDeformableAttnFunction_fp32 backward at the configs[2] self-attention shape (200 x 200 BEV, raster queries
    sampling around their own cell): the atomic-free kernels are taken (grad_value starts uninitialised), equal the fp32
    global-atomic kernel to rounding, and two runs give the SAME bits (the atomic kernel does not promise that)."""
    from fb_bev_amd import _capi
    from fb_bev_amd.backward_projection import const_tensor
    from fb_bev_amd.ms_deform_attn import MultiScaleDeformableAttnFunction_fp32 as F32
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    H = W = 200
    B, M, Dh, P = 2, 8, 10, 4
    ss = const_tensor([[H, W]], dev)
    ls = const_tensor([0], dev)
    assert _capi.msda_bwd_ws_bytes(B, H * W, M, Dh, 1, H * W, P, [[H, W]]) > 0
    value = torch.randn(B, H * W, M, Dh, generator=g).to(dev)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    ref = torch.stack([(xs.flatten() + 0.5) / W, (ys.flatten() + 0.5) / H], -1)
    off = (torch.rand(B, H * W, M, 1, P, 2, generator=g) - 0.5) * 8.0 / torch.tensor([W, H])
    loc = (ref[None, :, None, None, None, :] + off).contiguous().to(dev)
    w = torch.rand(B, H * W, M, 1, P, generator=g).softmax(-1).contiguous().to(dev)
    go = torch.randn(B, H * W, M * Dh, generator=g).to(dev)

    def run(level_hw):
        gv = torch.full_like(value, float('nan')) if level_hw else torch.zeros_like(value)
        gl, gw = torch.zeros_like(loc), torch.zeros_like(w)
        _capi.msda_bwd(value, ss, ls, loc, w, go, gv, gl, gw, level_hw=level_hw)
        return gv, gl, gw
    a = run([[H, W]])
    b = run(None)
    a2 = run([[H, W]])
    torch.cuda.synchronize()
    assert not torch.isnan(a[0]).any()
    assert (a[0] - b[0]).abs().max() <= 1e-5 * b[0].abs().max()
    assert torch.allclose(a[1], b[1], atol=1e-4, rtol=1e-4) and torch.allclose(a[2], b[2], atol=1e-5, rtol=1e-5)
    assert torch.equal(a[0], a2[0])
    # through the autograd Function: the host level shapes ride on the const tensor
    v = value.clone().requires_grad_()
    out = F32.apply(v, ss, ls, loc, w, 64)
    out.backward(go)
    assert torch.equal(v.grad, a[0])


@pytest.mark.gpu
def test_rows_linear_inference_route_and_its_weight_cache():
    """rows_linear.Linear without autograd on a GPU: the split-operand MFMA kernel (fbbev_rows_linear_x3) within 2e-5 of the
    fp32 GEMM, ReLU in the epilogue, and NO stale fragments: an in-place weight update, a storage swap and load_state_dict are
    each followed by the new weights' result."""
    from fb_bev_amd import rows_linear as RL
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(21)
    lin = RL.Linear(80, 128).to(dev)
    x = torch.randn(4, 10000, 80, generator=g).to(dev)

    def check(relu=False):
        with torch.no_grad():
            y = lin(x, relu=relu)
            ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
            ref = ref.relu() if relu else ref
        assert (y.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
        return y
    assert not RL.x3_ok(x, 80, 128)                         # autograd on: the vendor GEMM / split-K route
    with torch.no_grad():
        assert RL.x3_ok(x, 80, 128)                         # inference: the split-operand kernel
    y0 = check()
    check(relu=True)
    with torch.no_grad():
        lin.weight.mul_(-0.5)                               # in-place (optimizer step)
    y1 = check()
    assert not torch.allclose(y0, y1)
    lin.weight.data = torch.randn(128, 80, generator=g).to(dev) * 0.1          # storage swap
    check()
    other = torch.nn.Linear(80, 128).to(dev)
    lin.load_state_dict(other.state_dict())
    y3 = check()
    with torch.no_grad():
        assert (y3 - other(x)).abs().max() <= 2e-5 * y3.abs().max()
