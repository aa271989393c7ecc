"""CPU ORACLE for the backward projection (BEV -> image, depth-aware deformable cross-attention).
TEST INFRASTRUCTURE ONLY -- never imported by fb_bev_amd.

Functional restatement (weights passed as a flat dict keyed like the module tree's state_dict) of
  BackwardProjection.forward ............ backward_projection/backward_projection.py:84-133
  CustormLearnedPositionalEncoding ...... bevformer_utils/positional_encoding.py:38-60
  BEVFormer.forward ..................... bevformer_utils/bevformer.py:71-132
  bevformer_encoder.forward ............. bevformer_utils/bevformer_encoder.py:123-203 (+52-120 via oracle.py)
  BEVFormerEncoderLayer.forward ......... bevformer_utils/bevformer_encoder.py:250-377
  DA_SpatialCrossAttention.forward ...... bevformer_utils/spatial_cross_attention_depth.py:85-223
  DA_MSDeformableAttention.forward ...... :464-601, the CUDA branch :579-595 (NOT the :596-598 CPU branch,
                                          which silently drops the depth weighting -- SURVEY H7)
  mmcv MultiScaleDeformableAttention / FFN / LayerNorm (mmcv-full 1.5.2, external; restated).
Pins (tests/test_oracle_backward_projection.py, fixtures from tests/golden/make_golden.py):
  * DA_SpatialCrossAttention rebatch / scatter / count logic: da_sca_stub_inner.npz (real class, stub inner attention);
  * DA_MSDeformableAttention: da_msda_cpu_branch.npz (real class, CPU branch) and da_msda_cuda_branch.npz (real class
    driven through its CUDA branch :578-595 on the CPU, the mmcv op answered by oracle.msda_fwd);
  * mmcv MultiScaleDeformableAttention.forward: mmcv_msda_forward_trt_twin.npz -- the in-tree copy of that forward,
    multi_scale_deformable_attn_function.py:174-260 (MultiScaleDeformableAttentionTRT), run for real;
  * the MSDA op itself: tests/test_oracle_msda_ref.py (reference-tree twin of mmcv's bilinear device functions).
  mmcv's FFN / LayerNorm and the __init__ defaults of its attention class remain restatements.
The deformable sampling itself uses oracle.msda_grid_sample (F.grid_sample formulation).
"""
import torch
import torch.nn.functional as F

from . import oracle as O


def positional_encoding(P, pre, bs, h, w):
    x_embed = P[pre + 'col_embed.weight'][torch.arange(w)]
    y_embed = P[pre + 'row_embed.weight'][torch.arange(h)]
    pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)), dim=-1)
    return pos.permute(2, 0, 1).unsqueeze(0).repeat(bs, 1, 1, 1)


def _lin(P, name, x):
    return F.linear(x, P[name + '.weight'], P[name + '.bias'])


def mmcv_msda_self_attention(P, pre, query, query_pos, reference_points, spatial_shapes, level_start_index,
                             num_heads=8, num_levels=1, num_points=4):
    """mmcv MultiScaleDeformableAttention.forward with batch_first=True, value=None, identity=None
    (how bevformer_encoder.py:327-341 calls it)."""
    value = query
    identity = query
    query = query + query_pos
    bs, num_query, _ = query.shape
    num_value = value.shape[1]
    assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == num_value
    value = _lin(P, pre + 'value_proj', value).view(bs, num_value, num_heads, -1)
    so = _lin(P, pre + 'sampling_offsets', query).view(bs, num_query, num_heads, num_levels, num_points, 2)
    aw = _lin(P, pre + 'attention_weights', query).view(bs, num_query, num_heads, num_levels * num_points)
    aw = aw.softmax(-1).view(bs, num_query, num_heads, num_levels, num_points)
    norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + so / norm[None, None, None, :, None, :]
    out = O.msda_grid_sample(value, spatial_shapes, loc, aw)
    out = _lin(P, pre + 'output_proj', out)
    return out + identity


def da_msda(P, pre, query, value, reference_points, spatial_shapes, level_start_index, bev_query_depth,
            pred_img_depth, num_heads=8, num_levels=1, num_points=8, depth_weighting=True):
    """DA_MSDeformableAttention.forward (:464-601), batch_first=True, CUDA-branch semantics."""
    bs, num_query, _ = query.shape
    num_value = value.shape[1]
    value = _lin(P, pre + 'value_proj', value).view(bs, num_value, num_heads, -1)
    so = _lin(P, pre + 'sampling_offsets', query).view(bs, num_query, num_heads, num_levels, num_points, 2)
    aw = _lin(P, pre + 'attention_weights', query).view(bs, num_query, num_heads, num_levels * num_points)
    aw = aw.softmax(-1).view(bs, num_query, num_heads, num_levels, num_points)
    norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    _, _, Za, _ = reference_points.shape
    ref = reference_points[:, :, None, None, None, :, :]
    so = so / norm[None, None, None, :, None, :]
    so = so.view(bs, num_query, num_heads, num_levels, num_points // Za, Za, 2)
    loc = (ref + so).view(bs, num_query, num_heads, num_levels, num_points, 2)
    if not depth_weighting:   # the reference's non-CUDA branch (:596-598), only used to pin the fixture
        return O.msda_grid_sample(value, spatial_shapes, loc, aw)
    # :584-590 depth distribution sampled at each Z-anchor reference point (1 head x DC channels, weight 1)
    dref = reference_points.reshape(bs, num_query * Za, 1, 1, 1, 2)
    dsamp = O.msda_grid_sample(pred_img_depth.unsqueeze(2), spatial_shapes[0:1], dref,
                               torch.ones_like(dref[..., 0])).reshape(bs, num_query, Za, -1)
    dw = (dsamp * bev_query_depth).sum(-1)
    dw = dw.unsqueeze(2).repeat(1, 1, num_points // Za, 1).reshape(bs, num_query, num_points)
    aw = aw * dw[:, :, None, None, :]          # :592 -- no renormalisation
    return O.msda_grid_sample(value, spatial_shapes, loc, aw)


def da_spatial_cross_attention(P, pre, query, key, value, query_pos, reference_points_cam, per_cam_mask,
                               bev_query_depth, pred_img_depth, spatial_shapes, level_start_index, dbound,
                               num_cams=6, inner=None, return_slots=False, **attn_kw):
    """DA_SpatialCrossAttention.forward (:85-223), loops and all."""
    N, B, len_query, Z, _ = bev_query_depth.shape
    B, N, DC, H, W = pred_img_depth.shape
    bev_query_depth = bev_query_depth.permute(1, 0, 2, 3, 4)
    pred = pred_img_depth.view(B * N, DC, H, W).flatten(2).permute(0, 2, 1)
    inp_residual = query
    slots = torch.zeros_like(query)
    query = query + query_pos
    bs, num_query, E = query.size()
    D = reference_points_cam.size(3)
    indexes = [[] for _ in range(bs)]
    max_len = 0
    for j in range(bs):
        for i, m in enumerate(per_cam_mask):
            idx = m[j].sum(-1).nonzero().squeeze(-1)
            indexes[j].append(idx)
            max_len = max(max_len, len(idx))
    q_re = query.new_zeros([bs, num_cams, max_len, E])
    r_re = reference_points_cam.new_zeros([bs, num_cams, max_len, D, 2])
    d_re = reference_points_cam.new_zeros([bs, num_cams, max_len, D, 1])
    for j in range(bs):
        for i in range(num_cams):
            idx = indexes[j][i]
            q_re[j, i, :len(idx)] = query[j, idx]
            d_re[j, i, :len(idx)] = bev_query_depth[j, i, idx]
            r_re[j, i, :len(idx)] = reference_points_cam[i][j, idx]
    ncam, l, bs_, E_ = key.shape
    value = value.permute(2, 0, 1, 3).reshape(bs * num_cams, l, E)
    d_re = (d_re - dbound[0]) / dbound[2]
    d_re = torch.clip(torch.floor(d_re), 0, DC - 1).to(torch.long)
    onehot = F.one_hot(d_re.squeeze(-1), num_classes=DC)
    if inner is None:
        def inner(query, value, reference_points, spatial_shapes, level_start_index, bev_query_depth, pred_img_depth):
            return da_msda(P, pre + 'deformable_attention.', query, value, reference_points, spatial_shapes,
                           level_start_index, bev_query_depth, pred_img_depth, **attn_kw)
    out = inner(q_re.view(bs * num_cams, max_len, E), value, r_re.view(bs * num_cams, max_len, D, 2),
                spatial_shapes, level_start_index, onehot.view(bs * num_cams, max_len, D, DC),
                pred).view(bs, num_cams, max_len, E)
    for j in range(bs):
        for i in range(num_cams):
            idx = indexes[j][i]
            slots[j, idx] += out[j, i, :len(idx)]
    count = (per_cam_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1)
    slots = slots / torch.clamp(count, min=1.0)[..., None]
    if return_slots:
        return slots
    slots = _lin(P, pre + 'output_proj', slots)
    return slots + inp_residual


def ffn(P, pre, x):
    """mmcv FFN (num_fcs=2, ReLU, add_identity)."""
    h = F.relu(_lin(P, pre + 'layers.0.0', x))
    return x + _lin(P, pre + 'layers.1', h)


def layer_norm(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + 'weight'], P[pre + 'bias'])


def reference_points_2d(H, W, bs):
    """bevformer_encoder.py:78-89."""
    ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing='ij')
    ref_y = ref_y.reshape(-1)[None] / H
    ref_x = ref_x.reshape(-1)[None] / W
    return torch.stack((ref_x, ref_y), -1).repeat(bs, 1, 1).unsqueeze(2)


def backward_projection(P, mlvl_feats, lss_bev, cam_params, pred_img_depth, bev_h, bev_w, grid_config_bevformer,
                        final_dim, dbound, num_layers=1, num_heads=8, self_points=4, cross_points=8,
                        use_cams_embeds=False, inverse=torch.inverse):
    """BackwardProjection.forward -> (B, C, bev_h, bev_w). Weight keys follow the module tree."""
    bs, num_cam = mlvl_feats[0].shape[:2]
    bev_queries = P['bev_embedding.weight'].unsqueeze(1).repeat(1, bs, 1)
    bev_queries = bev_queries + lss_bev.flatten(2).permute(2, 0, 1)
    bev_pos = positional_encoding(P, 'positional_encoding.', bs, bev_h, bev_w)
    # BEVFormer.forward
    bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
    feats, shapes = [], []
    for feat in mlvl_feats:
        _, _, c, h, w = feat.shape
        f = feat.flatten(3).permute(1, 0, 3, 2)
        ce = P['transformer.cams_embeds'][:, None, None, :]
        f = f + (ce if use_cams_embeds else ce * 0)
        shapes.append((h, w))
        feats.append(f)
    feat_flatten = torch.cat(feats, 2).permute(0, 2, 1, 3)          # (num_cam, sum HW, bs, C)
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    # bevformer_encoder.forward
    ref_3d = O.reference_points_3d(grid_config_bevformer)
    ref_2d = reference_points_2d(bev_h, bev_w, bs)
    ref_cam, mask, qdepth = O.point_sampling(ref_3d, cam_params, final_dim, inverse=inverse)
    query = bev_queries.permute(1, 0, 2)
    pos = bev_pos.permute(1, 0, 2)
    ss_bev = torch.tensor([[bev_h, bev_w]])
    for lid in range(num_layers):
        pre = f'transformer.encoder.layers.{lid}.'
        query = mmcv_msda_self_attention(P, pre + 'attentions.0.', query, pos, ref_2d, ss_bev, torch.tensor([0]),
                                         num_heads=num_heads, num_levels=1, num_points=self_points)
        query = layer_norm(P, pre + 'norms.0.', query)
        query = da_spatial_cross_attention(P, pre + 'attentions.1.', query, feat_flatten, feat_flatten, pos, ref_cam,
                                           mask, qdepth, pred_img_depth, spatial_shapes, level_start_index, dbound,
                                           num_cams=num_cam, num_heads=num_heads, num_levels=len(shapes),
                                           num_points=cross_points)
        query = layer_norm(P, pre + 'norms.1.', query)
        query = ffn(P, pre + 'ffns.0.', query)
        query = layer_norm(P, pre + 'norms.2.', query)
    return query.permute(0, 2, 1).view(bs, -1, bev_h, bev_w).contiguous()
