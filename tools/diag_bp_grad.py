import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_backward_projection import _setup
from oracle import backward_projection_oracle as BO, oracle as O
dev = torch.device('cuda:0')
bev = 12
m, cfg, cam, feats, depth, lss, gcb = _setup(dev, B=1, num_levels=1, bev=bev, seed=3)
cam_g = [t.to(dev) for t in cam]
f_g = [f.to(dev).requires_grad_() for f in feats]; d_g = depth.to(dev).requires_grad_(); l_g = lss.to(dev).requires_grad_()
comp = m(f_g, None, lss_bev=l_g, cam_params=cam_g, pred_img_depth=d_g)
w = torch.randn(comp.shape, generator=torch.Generator().manual_seed(9))
(comp * w.to(dev)).sum().backward()
P = {k: v.detach().cpu().double().requires_grad_() for k, v in m.state_dict().items()}
f_c = [f.double().requires_grad_() for f in feats]; d_c = depth.double().requires_grad_(); l_c = lss.double().requires_grad_()
out_c = BO.backward_projection(P, f_c, l_c, cam, d_c, bev, bev, gcb, (256, 704), cfg['depth_bound'], inverse=O.inv3x3_closed_form)
(out_c * w.double()).sum().backward()
print('fwd max err', (comp.detach().cpu().double() - out_c).abs().max().item())
def rep(name, a, b):
    scale = b.abs().max().item() + 1e-12
    err = (a.double().cpu() - b).abs() / scale
    print(f'{name:70s} max_rel={err.max().item():.3e} frac>2e-3={(err > 2e-3).double().mean().item():.4f}')
    return err
e = rep('lss', l_g.grad, l_c.grad)
eq = e.amax(dim=1)[0]  # (bev,bev)
print((eq > 2e-3).int())
rep('feat', f_g[0].grad, f_c[0].grad); rep('depth', d_g.grad, d_c.grad)
for name, p in m.named_parameters():
    if p.grad is not None and P[name].grad is not None:
        rep(name, p.grad, P[name].grad)
