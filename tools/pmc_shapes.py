"""Dev tool: run a few representative engine launches (for rocprofv3 --pmc passes)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B = 64
def conv_case(cin, L, cout, ks, s, p, which):
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(B, cout, Lout, device=dev)
    with K.weight_cache():
        for _ in range(3):
            if which == 0: K.conv1d_fwd(x, w, None, s, p, act=1)
            elif which == 1: K.conv1d_bwd_data(dy, w, L, s, p)
            else: K.conv1d_bwd_weight(x, dy, ks, s, p)
    torch.cuda.synchronize()
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
for _ in range(3): K.gemm(0, a, b)
torch.cuda.synchronize()
conv_case(64, 4800, 128, 25, 4, 11, 0)     # audio l3 fwd
conv_case(32, 19200, 64, 25, 4, 11, 2)     # audio l2 bwd_weight
conv_case(256, 300, 512, 25, 4, 11, 1)     # audio l5 bwd_data
conv_case(128, 120, 128, 7, 1, 3, 0)       # TCN fwd
conv_case(128, 120, 128, 7, 1, 3, 2)       # TCN bwd_weight
