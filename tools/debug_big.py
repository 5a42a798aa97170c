import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from music2dance_amd import kernels
K = kernels.impl(); DEV = "cuda:0"
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double()
    return (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
g = torch.Generator().manual_seed(0)
for (B, cin, L, cout, ks, s, p) in [(64, 256, 300, 512, 25, 4, 11), (32, 256, 300, 512, 25, 4, 11), (64, 128, 1200, 256, 25, 4, 11), (32, 128, 1200, 256, 25, 4, 11), (16, 64, 4800, 128, 25, 4, 11), (128, 128, 120, 128, 7, 1, 3), (64, 128, 120, 128, 7, 1, 3), (128, 69, 120, 128, 25, 1, 12),
                                     (64, 69, 120, 128, 25, 1, 12), (128, 128, 120, 100, 120, 1, 0), (32, 128, 120, 128, 7, 1, 3)]:
    x = torch.randn(B, cin, L, generator=g); w = torch.randn(cout, cin, ks, generator=g) / math.sqrt(cin * ks)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(B, cout, Lout, generator=g); mask = torch.randn(B, cout, Lout, generator=g)
    m0 = (mask > 0).double()
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    out = F.conv1d(x64, w64, None, stride=s, padding=p)
    gx, gw = torch.autograd.grad(out, (x64, w64), dy.double() * m0)
    gx2, gw2 = torch.autograd.grad(F.conv1d(x64, w64, None, stride=s, padding=p), (x64, w64), dy.double())
    xd, wd, dyd, md = x.to(DEV), w.to(DEV), dy.to(DEV), mask.to(DEV)
    print((B, cin, L, cout, ks), "fwd %.1e" % rel(K.conv1d_fwd(xd, wd, None, s, p), out),
          "fwd_mask %.1e" % rel(K.conv1d_fwd(xd, wd, None, s, p, 0, 0.0, None, md, 0.0), out * m0),
          "bwd_data %.1e" % rel(K.conv1d_bwd_data(dyd, wd, L, s, p), gx2),
          "bwd_data_mask %.1e" % rel(K.conv1d_bwd_data(dyd, wd, L, s, p, md, 0.0), gx),
          "bwd_w %.1e" % rel(K.conv1d_bwd_weight(xd, dyd, ks, s, p), gw2),
          "bwd_w_mask %.1e" % rel(K.conv1d_bwd_weight(xd, dyd, ks, s, p, md, 0.0), gw),
          "chsum %.1e" % rel(K.channel_sums(dyd, md, 0.0), (dy.double() * m0).sum((0, 2))), flush=True)
