"""Dev tool: weight-gradient launches of the critic at their step sizes (audio branch 2B = 128 rows, pose branch 3B = 192)
under forced (tile, splits) plans. Needs a -DM2D_TUNING build (M2D_LIB=...)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
def timeit(fn, iters=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
CASES = [("audio_d.l2", 128, 32, 19200, 64, 25, 4, 11), ("audio_d.l3", 128, 64, 4800, 128, 25, 4, 11),
         ("audio_d.l4", 128, 128, 1200, 256, 25, 4, 11), ("audio_d.l5", 128, 256, 300, 512, 25, 4, 11),
         ("temporal.k7", 192, 128, 120, 128, 7, 1, 3), ("stick.conv1", 192, 69, 120, 128, 25, 1, 12)]
only = os.environ.get("CASE")
for name, b, cin, L, cout, ks, s, p in CASES:
    if only and only not in name: continue
    x = torch.randn(b, cin, L, device=dev); Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev)
    gf = 2.0 * b * Lout * cout * cin * ks / 1e6
    fn = lambda: K.conv1d_bwd_weight(x, dy, ks, s, p, with_bias=True, bias_from_sample=b // 2)
    os.environ.pop("M2D_PLAN", None)
    timeit(fn)
    res = [(timeit(fn), "auto")]
    for bm in (128, 64, 32):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
            os.environ["M2D_PLAN"] = "%d,%d" % (bm, sp)
            try:
                res.append((timeit(fn), (bm, sp)))
            except Exception as e:
                pass
    os.environ.pop("M2D_PLAN", None)
    auto = res[0][0]
    best = sorted(res[1:])[:6]
    print("%-12s auto %7.1f us %6.1f TF | " % (name, auto, gf / auto) + "  ".join("%s %.1f us %.1f TF" % (pl, t, gf / t) for t, pl in best), flush=True)
