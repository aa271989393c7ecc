"""Random DA cross-attention cases + the oracle's composite result, shared by the emulator tests (tests/test_emu_kernels.py)
and the GPU bound test of the fused backward (tests/test_gpu_backward_projection.py).  Test infrastructure."""
import torch


def da_case(seed=0, B=2, N=6, Q=70, Za=4, E=16, M=4, shapes=((5, 7), (3, 4)), P=8, DC=12, grad=False, extras=False):
    """Random DA cross-attention case + the oracle's composite result (slots before output_proj)."""
    from oracle import backward_projection_oracle as BO
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    ss = torch.tensor(shapes); ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    S_ = int((ss[:, 0] * ss[:, 1]).sum())
    H0, W0 = shapes[0]
    dbound = [2.0, 2.0 + DC, 1.0]
    Pm = {}
    for name, o in (('value_proj', E), ('sampling_offsets', M * L * P * 2), ('attention_weights', M * L * P)):
        Pm['a.deformable_attention.' + name + '.weight'] = torch.randn(o, E, generator=g) * 0.3
        Pm['a.deformable_attention.' + name + '.bias'] = torch.randn(o, generator=g) * 0.3
    Pm['a.output_proj.weight'] = torch.eye(E); Pm['a.output_proj.bias'] = torch.zeros(E)
    query = torch.randn(B, Q, E, generator=g); qpos = torch.randn(B, Q, E, generator=g)
    key = torch.randn(N, S_, B, E, generator=g)
    ref_cam = torch.rand(N, B, Q, Za, 2, generator=g) * 1.2 - 0.1
    mask = torch.rand(N, B, Q, Za, generator=g) < 0.2
    mask[2] = False
    qdepth = torch.rand(N, B, Q, Za, 1, generator=g) * (DC + 4.0)
    pred = torch.rand(B, N, DC, H0, W0, generator=g).softmax(2).contiguous()
    if grad:      # double-precision leaves for the autograd cross-check of the fused backward
        Pm = {k: v.double().requires_grad_() for k, v in Pm.items()}
        query, qpos, key, ref_cam, qdepth = (t.double() for t in (query, qpos, key, ref_cam, qdepth))
        pred = pred.double().requires_grad_()
        key.requires_grad_()
    exp = BO.da_spatial_cross_attention(Pm, 'a.', query, key, key, qpos, ref_cam, mask, qdepth, pred, ss, ls, dbound,
                                        num_cams=N, return_slots=True, num_heads=M, num_levels=L, num_points=P)
    # what the host hands to the fused kernel: camera-independent per-query projections
    import torch.nn.functional as F
    qq = query + qpos
    offsets = F.linear(qq, Pm['a.deformable_attention.sampling_offsets.weight'],
                       Pm['a.deformable_attention.sampling_offsets.bias']).view(B, Q, M, L, P, 2).contiguous()
    attn = F.linear(qq, Pm['a.deformable_attention.attention_weights.weight'],
                    Pm['a.deformable_attention.attention_weights.bias']).view(B, Q, M, L * P).softmax(-1)
    attn = attn.view(B, Q, M, L, P).contiguous()
    value = F.linear(key.permute(2, 0, 1, 3).reshape(B * N, S_, E), Pm['a.deformable_attention.value_proj.weight'],
                     Pm['a.deformable_attention.value_proj.bias']).view(B * N, S_, M, E // M).contiguous()
    args = (value, ss, ls, pred.view(B * N, DC, H0, W0), ref_cam.contiguous(), mask.contiguous(),
            qdepth.squeeze(-1).contiguous(), offsets, attn, dbound[0], dbound[2])
    if grad:
        return args, exp, dict(Pm=Pm, key=key, pred=pred)
    if extras:      # what the one-kernel entry (fbbev_da_cross_attn_fused) consumes instead of offsets / attn
        return args, exp, dict(Pm=Pm, query=query, qpos=qpos, key=key)
    return args, exp
