"""Per-layer timing of the conv engine at the phase-3 B=64 sizes. Dev tool."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels

K = kernels.impl()
dev = "cuda:0"
B = int(os.environ.get("B", 64))
N = B * 120
CASES = [
    ("stick.conv1", B, 69, 120, 128, 25, 1, 12),
    ("temporal.k7", B, 128, 120, 128, 7, 1, 3),
    ("stick.fconv", B, 128, 120, 100, 120, 1, 0),
    ("audio_d.l1", B, 1, 76800, 32, 25, 4, 11),
    ("audio_d.l2", B, 32, 19200, 64, 25, 4, 11),
    ("audio_d.l3", B, 64, 4800, 128, 25, 4, 11),
    ("audio_d.l4", B, 128, 1200, 256, 25, 4, 11),
    ("audio_d.l5", B, 256, 300, 512, 25, 4, 11),
    ("audio_d.l6", B, 512, 75, 100, 75, 1, 0),
    ("enc.c0", N, 1, 3200, 32, 250, 50, 124),
    ("enc.c1", N, 32, 64, 64, 4, 2, 1),
    ("enc.c2", N, 64, 32, 128, 4, 2, 1),
    ("enc.c3", N, 128, 16, 256, 4, 2, 1),
    ("enc.c4", N, 256, 8, 512, 4, 2, 1),
    ("enc.c5", N, 512, 4, 1024, 4, 2, 1),
    ("enc.c6", N, 1024, 2, 250, 2, 1, 0),
]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

for n in (4096,):
    a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
    for mode in (0, 1, 2):
        t = timeit(lambda: K.gemm(mode, a, b))
        print("gemm mode %d %d^3: %.3f ms %.1f TF/s" % (mode, n, t, 2.0 * n ** 3 / t / 1e9))
tot = {"fwd": 0, "bwd_data": 0, "bwd_weight": 0}
print("%-14s %9s | %8s %7s | %8s %7s | %8s %7s" % ("layer", "GF", "fwd ms", "TF/s", "bwdD ms", "TF/s", "bwdW ms", "TF/s"))
for name, b, cin, L, cout, ks, s, p in CASES:
    x = torch.randn(b, cin, L, device=dev)
    w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev)
    gf = 2.0 * b * Lout * cout * cin * ks / 1e9
    t1 = timeit(lambda: K.conv1d_fwd(x, w, bias, s, p, act=1))
    t2 = timeit(lambda: K.conv1d_bwd_data(dy, w, L, s, p))
    t3 = timeit(lambda: K.conv1d_bwd_weight(x, dy, ks, s, p))
    tot["fwd"] += t1; tot["bwd_data"] += t2; tot["bwd_weight"] += t3
    print("%-14s %9.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f" % (name, gf, t1, gf / t1, t2, gf / t2, t3, gf / t3))
print("totals ms", tot)
