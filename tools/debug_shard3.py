import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_full_size as F
import bench
from music2dance_amd import kernels, ops
K = kernels.impl()
DEV = "cuda:0"
gen, critic = bench.build_models(torch.device(DEV), 120)
blk = critic.stick_d.blocks[1]
g = torch.Generator().manual_seed(1)
for Bn in (128, 64):
    X = torch.randn(Bn, 128, 120, generator=g).to(DEV)
    R = torch.randn(Bn, 128, 120, generator=g).to(DEV)
    def run(x, r):
        x = x.clone().requires_grad_(True)
        blk.zero_grad(set_to_none=True)
        out = blk(x)
        (out * r).sum().backward()
        return x.grad.clone(), [p.grad.clone() for p in blk.parameters()], out.detach()
    gx, gp, out = run(X, R)
    h = Bn // 2
    gx1, gp1, out1 = run(X[:h].contiguous(), R[:h].contiguous())
    gx2, gp2, out2 = run(X[h:].contiguous(), R[h:].contiguous())
    print("B", Bn, "out", F.rel(out[:h], out1), "gx", F.rel(gx[:h], gx1), F.rel(gx[h:], gx2),
          "params", [("%.1e" % F.rel(a, b + c)) for a, b, c in zip(gp, gp1, gp2)])
    # direct kernel check on the same tensors: h1, h2 recomputed
    w1, b1, w2, b2 = blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias
    h1 = K.conv1d_fwd(X, w1, b1, 1, 3, 1)
    h2 = K.conv1d_fwd(h1, w2, b2, 1, 3, 1)
    d2 = K.conv1d_bwd_data(R, w2.detach(), 120, 1, 3, h2, 0.0)
    d2s = K.conv1d_bwd_data(R[:h].contiguous(), w2.detach(), 120, 1, 3, h2[:h].contiguous(), 0.0)
    print("   direct masked bwd_data full vs shard", F.rel(d2[:h], d2s))
    import torch.nn.functional as TF
    ref = torch.autograd.grad(TF.conv1d(h1.double().cpu().requires_grad_(True), w2.double().cpu(), None, padding=3), [], allow_unused=True) if False else None
    m = (h2 > 0).double().cpu()
    xin = h1.double().cpu().requires_grad_(True)
    o = TF.conv1d(xin, w2.detach().double().cpu(), None, padding=3)
    (gref,) = torch.autograd.grad(o, xin, R.double().cpu() * m)
    print("   direct masked bwd_data vs cpu", F.rel(d2, gref))
