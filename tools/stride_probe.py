"""Dev tool: what does the stride-4 activation gather cost? Same M, N, K with stride 4 and stride 1."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl(); dev = "cuda:0"
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
B = 64
for name, cin, cout, Lout in (("l2", 32, 64, 4800), ("l3", 64, 128, 1200), ("l4", 128, 256, 300), ("l5", 256, 512, 75)):
    for s, p in ((4, 11), (1, 12)):
        L = (Lout - 1) * s + 25 - 2 * p
        x = torch.randn(B, cin, L, device=dev); w = torch.randn(cout, cin, 25, device=dev) / math.sqrt(cin * 25)
        dy = torch.randn(B, cout, Lout, device=dev)
        gf = 2.0 * B * Lout * cout * cin * 25 / 1e9
        with K.weight_cache():
            t1 = timeit(lambda: K.conv1d_fwd(x, w, None, s, p, act=1))
            t2 = timeit(lambda: K.conv1d_bwd_weight(x, dy, 25, s, p))
            t3 = timeit(lambda: K.conv1d_bwd_data(dy, w, L, s, p))
        print("%s stride %d (L=%d): fwd %.1f us %.1f TF | bwdW %.1f us %.1f TF | bwdD %.1f us %.1f TF" % (
            name, s, L, 1e3 * t1, gf / t1, 1e3 * t2, gf / t2, 1e3 * t3, gf / t3), flush=True)
