"""CPU: the backward-projection module tree builds from the reference config block and exposes the
reference's parameter names (so reference checkpoints load); forward needs the GPU (no fallback)."""
import pytest
import torch

from fb_bev_amd import _capi, backward_projection as BP, configs


def test_builds_from_config_with_reference_state_dict_keys():
    cfg = configs.fbocc_r50()
    m = BP.build(cfg['backward_projection'])
    keys = set(m.state_dict().keys())
    pre = 'transformer.encoder.layers.0.'
    expect = {'bev_embedding.weight', 'positional_encoding.row_embed.weight', 'positional_encoding.col_embed.weight',
              'transformer.cams_embeds'}
    for n in ('sampling_offsets', 'attention_weights', 'value_proj', 'output_proj'):
        expect |= {f'{pre}attentions.0.{n}.weight', f'{pre}attentions.0.{n}.bias'}
    for n in ('sampling_offsets', 'attention_weights', 'value_proj'):
        expect |= {f'{pre}attentions.1.deformable_attention.{n}.weight', f'{pre}attentions.1.deformable_attention.{n}.bias'}
    expect |= {f'{pre}attentions.1.output_proj.weight', f'{pre}attentions.1.output_proj.bias'}
    expect |= {f'{pre}ffns.0.layers.0.0.weight', f'{pre}ffns.0.layers.0.0.bias', f'{pre}ffns.0.layers.1.weight',
               f'{pre}ffns.0.layers.1.bias'}
    for i in range(3):
        expect |= {f'{pre}norms.{i}.weight', f'{pre}norms.{i}.bias'}
    assert keys == expect
    sd = m.state_dict()
    assert sd[f'{pre}attentions.0.sampling_offsets.weight'].shape == (8 * 1 * 4 * 2, 80)      # mmcv defaults: 8 heads, 4 pts
    assert sd[f'{pre}attentions.1.deformable_attention.sampling_offsets.weight'].shape == (8 * 1 * 8 * 2, 80)
    assert sd[f'{pre}ffns.0.layers.0.0.weight'].shape == (320, 80)
    assert sd['bev_embedding.weight'].shape == (10000, 80)
    # ring-pattern initial offsets (spatial_cross_attention_depth.py:442-458): point i of an anchor scaled by i+1
    b = sd[f'{pre}attentions.1.deformable_attention.sampling_offsets.bias'].view(8, 1, 2, 4, 2)
    assert torch.allclose(b[:, :, 1], 2 * b[:, :, 0])


def test_forward_refuses_cpu_tensors():
    cfg = configs.fbocc_r50(bev_h=4, bev_w=4)
    att = BP.build(dict(type='MultiScaleDeformableAttention', embed_dims=80, num_levels=1, batch_first=True))
    q = torch.zeros(1, 16, 80)
    with pytest.raises(_capi.FbbevError):
        att(q, query_pos=q, reference_points=torch.zeros(1, 16, 1, 2), spatial_shapes=torch.tensor([[4, 4]]),
            level_start_index=torch.tensor([0]))


def test_history_fusion_module_builds_with_detector_key_names():
    """fbocc.py:111-127: the two Sequentials keep their names so a detector checkpoint loads unchanged."""
    import pytest
    import torch
    from fb_bev_amd import _capi
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    m = TemporalHistoryFusion([0.8, 0.8, 0.8], [-39.6, -39.6, -0.6], single_bev_num_channels=80, history_cat_num=16)
    keys = set(m.state_dict().keys())
    assert {'history_keyframe_time_conv.0.weight', 'history_keyframe_time_conv.1.running_mean',
            'history_keyframe_cat_conv.0.weight', 'history_keyframe_cat_conv.1.running_var'} <= keys
    assert tuple(m.history_keyframe_time_conv[0].weight.shape) == (80, 81, 1, 1, 1)
    assert tuple(m.history_keyframe_cat_conv[0].weight.shape) == (80, 80 * 17, 1, 1, 1)
    assert m.lower == pytest.approx([-40.0, -40.0, -1.0])
    with pytest.raises(_capi.FbbevError):      # no CPU fallback
        m.fuse_history(torch.zeros(1, 80, 4, 4, 2), [dict(sequence_group_idx=0, start_of_sequence=True,
                                                          curr_to_prev_ego_rt=torch.eye(4))], torch.eye(3)[None])


def test_head_minor_projection_is_a_row_permutation_of_the_reference_projection():
    """DA_MSDeformableAttention.project_head_minor == project with (M,L,P) -> (L,P,M) transposed outputs."""
    import torch
    torch.manual_seed(0)
    da = BP.build(dict(type='DA_MSDeformableAttention', embed_dims=80, num_points=8, num_levels=4))
    with torch.no_grad():
        da.sampling_offsets.weight.normal_(0, 0.1)
        da.attention_weights.weight.normal_(0, 0.1); da.attention_weights.bias.normal_(0, 0.1)
    q = torch.randn(2, 37, 80)
    so, aw = da.project(q)
    so2, aw2 = da.project_head_minor(q)
    assert so2.shape == (2, 37, 4, 8, 8, 2) and aw2.shape == aw.shape
    assert torch.allclose(so2, so.permute(0, 1, 3, 4, 2, 5), atol=1e-6)
    assert torch.allclose(aw2, aw, atol=1e-6)


def test_fused_tail_is_refused_while_the_ffn_has_a_live_dropout():
    """the cross-attention tail + FFN one-kernel route (fbbev_rows_tail_ffn_x3) has no dropout; FB-OCC variants
    with ffn_dropout > 0 under model.train() + torch.no_grad() must keep the FFN's own Dropout layers -- fused_tail_spec answers None
    then, as FFN.forward itself keeps `self.layers`.  In eval() mode, or with p = 0 in train() mode, the route stays available
    (the spec itself needs GPU weights: only the gate is checked here)."""
    import torch
    from fb_bev_amd import backward_projection as BP
    ffn = BP.FFN(embed_dims=80, feedforward_channels=320, ffn_drop=0.1)
    ffn.train()
    assert ffn._has_live_dropout() and ffn.fused_tail_spec(80) is None
    ffn0 = BP.FFN(embed_dims=80, feedforward_channels=320, ffn_drop=0.0)
    ffn0.train()
    assert not ffn0._has_live_dropout()
    # the training route's gate refuses a layer with a live dropout as well
    from fb_bev_amd import configs, train_path as TP
    cfg = configs.fbocc_r50(bev_h=16, bev_w=16)
    cfg['backward_projection']['transformer']['encoder']['transformerlayers']['ffn_dropout'] = 0.1
    m = BP.build(cfg['backward_projection']).train()
    layer = m.transformer.encoder.layers[0]
    assert layer.ffns[0]._has_live_dropout()
    q = torch.zeros(1, 256, 80)
    assert not TP.layer_supported(layer, q, q, torch.zeros(6, 704, 80), None, torch.zeros(6, 1, 256, 4, 2), None, 16, 16, None)
