"""GPU checks of the assembled FB-OCC detector: the whole chain image -> occupancy runs through the HIP
path, the fused inference route equals the composite (autograd) route, the sync-free losses agree with their CPU
evaluation (which is pinned on the reference fixture, tests/test_occ_modules.py), and a training step neither
synchronises with the host nor leaves a block without gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _small_model(dev, execution=None, neck_channels=32):
    from fb_bev_amd import configs
    from fb_bev_amd.fbocc import FBOCC
    bev, C = 20, 80
    grid = {'x': [-40, 40, 4.0], 'y': [-40, 40, 4.0], 'z': [-1, 5.4, 0.8], 'depth': [2.0, 42.0, 0.5]}     # 20x20x8
    gcb = {'x': [-40, 40, 4.0], 'y': [-40, 40, 4.0], 'z': [-1, 5.4, 1.6]}
    blocks = configs.fbocc_r50(bev_h=bev, bev_w=bev, numC_Trans=C, grid_config=grid, grid_config_bevformer=gcb)
    pcr = [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4]
    cfg = dict(
        use_depth_supervision=True, fix_void=True, do_history=True, history_cat_num=2, single_bev_num_channels=C, readd=True,
        img_backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(2, 3), norm_eval=False, base_channels=8),
        img_neck=dict(type='CustomFPN', in_channels=[32, 64], out_channels=24, num_outs=1, start_level=0, out_ids=[0]),
        depth_net=dict(type='CM_DepthNet', in_channels=24, context_channels=C, downsample=16, grid_config=grid,
                       depth_channels=80, mid_channels=32, loss_depth_weight=1., use_dcn=False),
        forward_projection=blocks['forward_projection'], backward_projection=blocks['backward_projection'],
        img_bev_encoder_backbone=dict(type='CustomResNet3D', depth=18, block_strides=[1, 2, 2], n_input_channels=C,
                                      block_inplanes=[16, 32, 64], out_indices=(0, 1, 2), norm_cfg=dict(type='SyncBN')),
        img_bev_encoder_neck=dict(type='FPN3D', in_channels=[16, 32, 64], out_channels=neck_channels, norm_cfg=dict(type='SyncBN')),
        occupancy_head=dict(type='OccHead', use_focal_loss=True, norm_cfg=dict(type='SyncBN'), soft_weights=True,
                            final_occ_size=[40, 40, 16], empty_idx=18, num_level=3, in_channels=[neck_channels] * 3, out_channel=19,
                            point_cloud_range=pcr))
    torch.manual_seed(0)
    m = FBOCC(**cfg, execution=execution)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if 'sampling_offsets.weight' in name or 'attention_weights.weight' in name:
                p.normal_(0, 0.05)
    return m.to(dev)


def _inputs(dev, B, seed=0):
    from fb_bev_amd import synthetic as S
    pc = S.CONFIGS['REF']
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=seed, bda_aug=True)]
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.randn(B, 6, 3, 256, 704, generator=g).to(dev)
    gt_occ = torch.randint(1, 19, (B, 40, 40, 16), generator=g)
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.5] = 18
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.3] = 255
    gt_depth = torch.rand(B, 6, 256, 704, generator=g) * 40 + 2
    gt_depth[torch.rand(gt_depth.shape, generator=g) > 0.05] = 0

    def metas(first):
        return [dict(sequence_group_idx=b, start_of_sequence=first, curr_to_prev_ego_rt=torch.eye(4), index=b) for b in range(B)]
    return [img] + cam, metas, gt_occ.to(dev), gt_depth.to(dev)


def test_inference_route_equals_autograd_route_and_formats(dev):
    m = _small_model(dev).eval()
    img_inputs, metas, _, _ = _inputs(dev, 2)
    with torch.no_grad():                                    # fused / write-once kernels
        r0 = m.extract_feat(None, img_inputs, metas(True))
        logits0 = m.occupancy_head(r0['img_bev_feat'])['output_voxels'][0]
    m.reset_history()
    img_g = [img_inputs[0].clone().requires_grad_()] + img_inputs[1:]
    r1 = m.extract_feat(None, img_g, metas(True))            # grad enabled -> composite autograd path of every stage
    logits1 = m.occupancy_head(r1['img_bev_feat'])['output_voxels'][0]
    assert logits0.shape == (2, 19, 40, 40, 16)
    assert torch.allclose(logits0, logits1, atol=2e-3, rtol=2e-3), (logits0 - logits1).abs().max()
    assert (logits0 - logits1).abs().median() < 1e-5
    m.reset_history()
    with torch.no_grad():
        ids = m.predict_occupancy(img_inputs, metas(True))
        m.reset_history()
        one = [t[:1] for t in img_inputs]
        res = m(return_loss=False, img_inputs=[one], img_metas=[metas(True)[:1]])
    assert ids.shape == (2, 40, 40, 16) and int(ids.max()) <= 17 and int(ids.min()) >= 0
    assert res[0]['pred_occupancy'].shape == (40, 40, 16)
    assert (torch.from_numpy(res[0]['pred_occupancy']).to(dev) == ids[0]).float().mean() > 0.99


@pytest.mark.parametrize('execution,tol', [(dict(history_ring='voxel_major'), 1e-4),
                                           (dict(history_dtype='f16', history_compute='bf16', history_ring='voxel_major'), 5e-2)])
def test_detector_history_execution_knobs(dev, execution, tol):
    """The history knobs of the detector's `execution` block through a 3-frame sequence with ego motion: the voxel-major
    ring alone is the default detector to fp32 rounding (same elements, fp32 convolutions summed in another K order); with
    the fp16 ring and the bf16-MFMA convolutions the logits stay within the stated reduced-precision band and the predicted
    classes agree on > 97 % of the voxels."""
    base = _small_model(dev).eval()
    m = _small_model(dev, execution).eval()
    m.load_state_dict(base.state_dict())
    hist = m._path[1]
    assert hist._voxel_major()
    img_inputs, metas, _, _ = _inputs(dev, 2)
    ego = torch.eye(4); ego[0, 3] = 1.5; ego[1, 3] = -0.7
    for i in range(3):
        mt = [dict(d, curr_to_prev_ego_rt=ego) for d in metas(i == 0)]
        frame = [img_inputs[0] + 0.1 * i] + img_inputs[1:]
        with torch.no_grad():
            l0 = base.occupancy_head(base.extract_feat(None, frame, mt)['img_bev_feat'])['output_voxels'][0]
            l1 = m.occupancy_head(m.extract_feat(None, frame, mt)['img_bev_feat'])['output_voxels'][0]
        scale = l0.abs().max().item()
        assert (l0 - l1).abs().max().item() <= tol * scale, (i, (l0 - l1).abs().max().item(), scale)
        assert (l0.argmax(1) == l1.argmax(1)).float().mean().item() > 0.97
        assert hist.history_bev.dim() == 4 and hist.history_bev.dtype == hist.history_dtype
        h0 = base._path[1].history_as_reference()
        # the stored frames do not depend on the fused output: identical elements, or fp16 roundings of them (one per re-sampling)
        assert torch.allclose(hist.history_as_reference(), h0, rtol=0, atol=(2e-3 if 'history_dtype' in execution else 0.0) * h0.abs().max().item())


def test_losses_on_gpu_equal_cpu_evaluation_and_do_not_sync(dev):
    from fb_bev_amd import occ_loss as L
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 19, 200, 200, 4, generator=g)
    gt = torch.randint(1, 19, (2, 200, 200, 4), generator=g)
    gt[torch.rand(gt.shape, generator=g) < 0.5] = 18
    gt[torch.rand(gt.shape, generator=g) < 0.2] = 255
    gt[gt == 5] = 18
    cw = L.class_weights(19).float()
    focal = L.CustomFocalLoss()
    fns = {'focal': lambda x, t, w, f: f(x, t, w, ignore_index=255),
           'ce': lambda x, t, w, f: L.CE_ssc_loss(x, t, w, ignore_index=255),
           'sem': lambda x, t, w, f: L.sem_scal_loss(x, t),
           'geo': lambda x, t, w, f: L.geo_scal_loss(x, t, non_empty_idx=18),
           'lovasz': lambda x, t, w, f: L.lovasz_softmax(torch.softmax(x, 1), t, ignore=255)}
    cpu = {k: float(fn(logits, gt, cw, focal)) for k, fn in fns.items()}
    lg, tg, wg, fg = logits.to(dev).requires_grad_(), gt.to(dev), cw.to(dev), focal.to(dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        vals = {k: fn(lg, tg, wg, fg) for k, fn in fns.items()}
        sum(vals.values()).backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    for k, v in vals.items():
        assert abs(float(v) - cpu[k]) <= 1e-4 * abs(cpu[k]) + 1e-6, (k, float(v), cpu[k])
    assert torch.isfinite(lg.grad).all() and lg.grad.abs().sum() > 0


# voxel_dtype='bf16' is an inference-only setting (bev_encoder._low_precision_ok): under autograd that stack runs fp32
@pytest.mark.parametrize('execution', [None, dict(img_dtype='bf16', depth_dtype='bf16'), dict(head_dtype='bf16')])
def test_training_step_backpropagates_everywhere_without_host_sync(dev, execution):
    m = _small_model(dev, execution).train()
    img_inputs, metas, gt_occ, gt_depth = _inputs(dev, 2, seed=3)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-2)

    def step(first):
        opt.zero_grad(set_to_none=True)
        losses = m(return_loss=True, img_inputs=img_inputs, img_metas=metas(first), gt_occupancy=gt_occ, gt_depth=gt_depth)
        total = m.parse_losses(losses)
        total.backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=5, norm_type=2)
        return total, losses

    total, losses = step(True)                               # allocations, workspace caches, first-frame history
    assert set(losses) == {'loss_voxel_ce_c_0', 'loss_voxel_sem_scal_c_0', 'loss_voxel_geo_scal_c_0', 'loss_voxel_lovasz_c_0',
                           'loss_depth'}
    assert torch.isfinite(total)
    missing = [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:5]
    opt.step()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        total2, _ = step(False)
        opt.step()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert torch.isfinite(total2)


def test_baseline_config3_training_step_at_shipped_size(dev):
    """BASELINE configs[3]: ONE forward_train + backward of the WHOLE shipped FB-OCC R50 config (6x256x704 in, 100x100x8
    grid, 16-frame history) at the per-GPU batch of the 8x4 = 32 global batch, on the route bench.py --mode train times
    (3-D stacks on fbbev_conv3d_*).  Checks: the five losses are finite and equal their CPU re-evaluation from the
    GPU logits / depth distribution (the loss functions are pinned on the reference fixture, tests/test_occ_modules.py);
    every parameter of every block receives a finite gradient; the flat gradient buckets (shard.GradBuckets: what
    the DDP step all-reduces) hold exactly those gradients; no host synchronisation in the second step."""
    import json
    import os
    from fb_bev_amd import shard, synthetic as S
    from fb_bev_amd.fbocc import FBOCC
    cfg = dict(json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'fbocc_config_path_blocks.json')))
               ['fbocc-r50-cbgs_depth_16f_16x4_20e.py']['model'])
    cfg.pop('type')
    torch.manual_seed(0)
    m = FBOCC(**cfg, execution=dict(with_cp=False, mfma_conv3d_train=True)).to(dev).train()
    B = 4
    pc = S.CONFIGS['REF']
    g = torch.Generator().manual_seed(11)
    cam = [t.to(dev) for t in S.camera_rig(pc, B, seed=11, bda_aug=True)]
    img = torch.randn(B, 6, 3, 256, 704, generator=g).to(dev)
    gt_depth = torch.rand(B, 6, 256, 704, generator=g) * 40 + 2
    gt_depth[torch.rand(gt_depth.shape, generator=g) > 0.03] = 0
    gt_occ = torch.randint(1, 19, (B, 200, 200, 16), generator=g)
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.6] = 18
    gt_occ[torch.rand(gt_occ.shape, generator=g) < 0.4] = 255
    gt_occ, gt_depth = gt_occ.to(dev), gt_depth.to(dev)

    def metas(first):
        return [dict(sequence_group_idx=b, start_of_sequence=first, curr_to_prev_ego_rt=torch.eye(4), index=b) for b in range(B)]
    model, buckets = shard.prepare_ddp(m, sync_bn=False)
    captured = {}
    head_loss = m.occupancy_head.loss                       # forward_train calls forward() / loss() directly: wrap loss()

    def capture_loss(**kw):
        captured['occ'] = [v.detach().float().cpu() for v in kw['output_voxels']]
        return head_loss(**kw)
    m.occupancy_head.loss = capture_loss
    h2 = m.depth_net.register_forward_hook(lambda mod, i, o: captured.__setitem__('depth', o))
    buckets.zero_grad()
    losses = m(return_loss=True, img_inputs=[img] + cam, img_metas=metas(True), gt_occupancy=gt_occ, gt_depth=gt_depth)
    h2.remove()
    m.occupancy_head.loss = head_loss
    total = m.parse_losses(losses)
    total.backward()
    buckets.finish()
    assert set(losses) == {'loss_voxel_ce_c_0', 'loss_voxel_sem_scal_c_0', 'loss_voxel_geo_scal_c_0', 'loss_voxel_lovasz_c_0',
                           'loss_depth'}
    assert all(torch.isfinite(v).all() for v in losses.values()) and torch.isfinite(total)
    # losses re-evaluated on the CPU from the GPU's own logits / depth distribution
    head_cpu = type(m.occupancy_head)(**{k: v for k, v in cfg['occupancy_head'].items() if k != 'type'})
    cpu_losses = head_cpu.loss(output_voxels=captured['occ'], target_voxels=gt_occ.cpu())
    for k, v in cpu_losses.items():
        assert abs(float(losses[k]) - float(v)) <= 2e-4 * abs(float(v)) + 1e-5, (k, float(losses[k]), float(v))
    import copy
    depth_cpu = copy.deepcopy(m.depth_net).cpu().get_depth_loss(gt_depth.cpu(), captured['depth'][1].detach().float().cpu())
    assert abs(float(losses['loss_depth']) - float(depth_cpu['loss_depth'])) <= 2e-4 * abs(float(depth_cpu['loss_depth'])) + 1e-5
    # every parameter has a finite gradient living in the flat buckets, every block a non-zero gradient norm
    norms = {}
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
            norms[n.split('.')[0]] = norms.get(n.split('.')[0], 0.0) + float(p.grad.double().pow(2).sum())
    assert set(norms) == {'img_backbone', 'img_neck', 'depth_net', 'backward_projection', 'history_keyframe_time_conv',
                          'history_keyframe_cat_conv', 'img_bev_encoder_backbone', 'img_bev_encoder_neck', 'occupancy_head'}
    assert all(v > 0 for v in norms.values()), norms
    flat = sum(float(f.double().pow(2).sum()) for f in buckets._flat)
    assert abs(flat - sum(norms.values())) <= 1e-6 * flat          # the buckets ARE the gradients (views, no copies)
    assert buckets.nbytes == 4 * sum(p.numel() for p in buckets.params) and len(buckets.buckets) >= 4
    print('configs[3] B=4 step: loss', float(total), 'grad norm per block', {k: round(v ** 0.5, 4) for k, v in norms.items()})
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        buckets.zero_grad()
        m.parse_losses(m(return_loss=True, img_inputs=[img] + cam, img_metas=metas(False), gt_occupancy=gt_occ,
                         gt_depth=gt_depth)).backward()
        buckets.finish()
    finally:
        torch.cuda.set_sync_debug_mode('default')
