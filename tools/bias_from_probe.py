"""Dev probe: conv1d_bwd_weight(bias_from_sample) at the critic-step's sizes vs an fp64 sum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl(); dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (R, B0, C, L, ks, pad, masked) in ((192, 64, 128, 120, 7, 3, False), (192, 64, 128, 120, 7, 3, True), (12, 4, 128, 120, 7, 3, True),
                                        (192, 64, 69, 120, 25, 12, False), (128, 64, 32, 4800, 25, 11, False)):
    stride = 4 if L > 1000 else 1
    Lout = (L + 2 * pad - ks) // stride + 1
    x = torch.randn(R, C, L, generator=g).to(dev)
    dy = torch.randn(R, 128 if L <= 1000 else 64, Lout, generator=g).to(dev)
    m = torch.randn(dy.shape, generator=g).to(dev) if masked else None
    dw, db = K.conv1d_bwd_weight(x, dy, ks, stride, pad, dy_mask=m, with_bias=True, bias_from_sample=B0)
    dw0, db0 = K.conv1d_bwd_weight(x, dy, ks, stride, pad, dy_mask=m, with_bias=True)
    h = dy.double() * ((m > 0).double() if masked else 1.0)
    ref = h[B0:].sum((0, 2)); ref0 = h.sum((0, 2))
    print(R, B0, C, L, ks, masked, "from: %.2e" % ((db.double() - ref).abs().max() / ref.abs().max()).item(),
          "all: %.2e" % ((db0.double() - ref0).abs().max() / ref0.abs().max()).item(),
          "dw same:", torch.equal(dw, dw0), flush=True)
