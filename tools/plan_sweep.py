"""Dev tool: time one conv shape under forced (bm, splits) plans. Needs a -DM2D_TUNING build."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

CASES = [("temporal.k7 B64", 64, 128, 120, 128, 7, 1, 3), ("temporal.k7 B128", 128, 128, 120, 128, 7, 1, 3),
         ("stick.conv1 B64", 64, 69, 120, 128, 25, 1, 12), ("audio_d.l5", 64, 256, 300, 512, 25, 4, 11),
         ("audio_d.l2", 64, 32, 19200, 64, 25, 4, 11)]
PLANS = [None, (128, 1), (64, 1), (32, 1), (32, 2), (32, 3), (32, 4), (64, 2), (64, 3), (64, 4), (128, 2), (128, 4), (128, 7)]
for name, b, cin, L, cout, ks, s, p in CASES:
    x = torch.randn(b, cin, L, device=dev); w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev); Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev); gf = 2.0 * b * Lout * cout * cin * ks / 1e9
    print("==", name, "GF %.2f" % gf)
    for pl in PLANS:
        if pl is None: os.environ.pop("M2D_PLAN", None)
        else: os.environ["M2D_PLAN"] = "%d,%d" % pl
        t1 = timeit(lambda: K.conv1d_fwd(x, w, bias, s, p, act=1))
        t2 = timeit(lambda: K.conv1d_bwd_data(dy, w, L, s, p))
        t3 = 0.0
        if pl is None or pl[1] > 1 or name.startswith("temporal") and False:
            t3 = timeit(lambda: K.conv1d_bwd_weight(x, dy, ks, s, p))
        print("%-10s fwd %7.1f us %5.1f TF | bwdD %7.1f us %5.1f TF | bwdW %7.1f us %5.1f TF" % (
            pl, 1e3 * t1, gf / t1, 1e3 * t2, gf / t2, 1e3 * t3, gf / t3 if t3 else 0), flush=True)
