"""Dev tool: forward / backward-data / backward-weight time of the audio-critic and first encoder convs at B = 64
(set M2D_LIB to compare two builds on one box)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B = 64
CASES = [("audio_d.l2", B, 32, 19200, 64, 25, 4, 11), ("audio_d.l3", B, 64, 4800, 128, 25, 4, 11),
         ("audio_d.l4", B, 128, 1200, 256, 25, 4, 11), ("audio_d.l5", B, 256, 300, 512, 25, 4, 11),
         ("enc.c1", B * 120, 32, 64, 64, 4, 2, 1), ("enc.c2", B * 120, 64, 32, 128, 4, 2, 1), ("enc.c5", B * 120, 512, 4, 1024, 4, 2, 1),
         ("temporal.k7", 2 * B, 128, 120, 128, 7, 1, 3)]
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for name, b, cin, L, cout, ks, s, p in CASES:
    x = torch.randn(b, cin, L, device=dev); w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev); Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev)
    with K.weight_cache():
        t = [timeit(lambda: K.conv1d_fwd(x, w, bias, s, p, act=1)), timeit(lambda: K.conv1d_bwd_data(dy, w, L, s, p, out_mask=x, out_mask_slope=0.0)),
             timeit(lambda: K.conv1d_bwd_weight(x, dy, ks, s, p))]
    print("%-12s fwd %7.1f us  bwdD(masked out) %7.1f us  bwdW %7.1f us" % (name, *t), flush=True)
