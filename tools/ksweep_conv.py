"""Dev tool: fixed-cost (prologue + epilogue + launch) vs per-chunk cost of a conv launch:
time against the channel count at fixed output size."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
B, L, Cout, ks, s, p = 64, 4800, 128, 25, 4, 11
for cin in (16, 32, 64, 128, 256):
    x = torch.randn(B, cin, L, device=dev); w = torch.randn(Cout, cin, ks, device=dev) * 0.02
    with K.weight_cache():
        t = timeit(lambda: K.conv1d_fwd(x, w, None, s, p, act=1))
    gf = 2.0 * B * 1200 * Cout * cin * ks / 1e9
    print("fwd  Cin %4d chunks %4d  %7.1f us  %6.1f TF/s" % (cin, 25 * cin // 16, t, gf / t * 1e3))
for cout in (32, 64, 128, 256):
    x = torch.randn(B, 64, L, device=dev); dy = torch.randn(B, cout, 1200, device=dev); w = torch.randn(cout, 64, ks, device=dev) * 0.02
    with K.weight_cache():
        t = timeit(lambda: K.conv1d_bwd_data(dy, w, L, s, p))
    gf = 2.0 * B * 1200 * cout * 64 * ks / 1e9
    print("bwdD Cout %4d              %7.1f us  %6.1f TF/s" % (cout, t, gf / t * 1e3))
