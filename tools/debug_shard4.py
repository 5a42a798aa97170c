import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as TF
import tests.test_gpu_full_size as F
import bench
from music2dance_amd import kernels, ops
K = kernels.impl()
DEV = "cuda:0"
gen, critic = bench.build_models(torch.device(DEV), 120)
conv = critic.stick_d.blocks[1].conv2
g = torch.Generator().manual_seed(1)
for Bn in (64, 64, 128, 64, 32, 48, 80):
    X = torch.randn(Bn, 128, 120, generator=g).to(DEV)
    R = torch.randn(Bn, 128, 120, generator=g).to(DEV)
    x = X.clone().requires_grad_(True)
    conv.zero_grad(set_to_none=True)
    y = conv(x, act=1)
    (y * R).sum().backward()
    # cpu reference
    xc = X.double().cpu().requires_grad_(True); wc = conv.weight.detach().double().cpu().requires_grad_(True); bc = conv.bias.detach().double().cpu().requires_grad_(True)
    yc = TF.relu(TF.conv1d(xc, wc, bc, padding=3))
    gxc, gwc, gbc = torch.autograd.grad((yc * R.double().cpu()).sum(), (xc, wc, bc))
    print("B", Bn, "y", F.rel(y, yc), "gx", F.rel(x.grad, gxc), "gw", F.rel(conv.weight.grad, gwc), "gb", F.rel(conv.bias.grad, gbc))
    d = K.conv1d_bwd_data(R, conv.weight.detach(), 120, 1, 3, y.detach(), 0.0)
    dw = K.conv1d_bwd_weight(X, R, 7, 1, 3, y.detach(), 0.0)
    mm = ((y.detach().cpu() > 0) != (yc.detach() > 0))
    pre = TF.conv1d(xc, wc, bc, padding=3).detach()
    print("   mask mismatches", int(mm.sum()), "of", mm.numel(), "max |pre| at mismatch", float(pre[mm].abs().max()) if mm.any() else 0.0,
          "y==0 frac", float((y == 0).float().mean()), "min positive y", float(y[y > 0].min()))
    ygpu_mask = (y.detach() > 0).float()
    gx_t = torch.nn.grad.conv1d_input(X.shape, conv.weight.detach(), R * ygpu_mask, padding=3)
    print("   torch-gpu gx vs cpu", F.rel(gx_t, gxc), " mine vs torch-gpu", F.rel(x.grad, gx_t))
    print("   direct: gx", F.rel(d, gxc), "gw", F.rel(dw, gwc), "autograd-vs-direct gx", F.rel(x.grad, d))
