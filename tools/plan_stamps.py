"""Dev tool: one conv / GEMM shape under forced (bm, splits) plans, with the per-workgroup phase stamps of each.
Needs a -DM2D_STAMP -DM2D_TUNING build (M2D_LIB=<that .so>). CASE=name selects (default: the TCN critic's 3B k7 layer)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music2dance_amd import kernels, _lib

K = kernels.impl()
L = _lib.lib()
dev = "cuda:0"
buf = (ctypes.c_ulonglong * (8192 * 8))()
CASES = {"tcn3b": (192, 128, 120, 128, 7, 1, 3), "tcn1b": (64, 128, 120, 128, 7, 1, 3), "stick1": (192, 69, 120, 128, 25, 1, 12),
         "l3": (64, 64, 4800, 128, 25, 4, 11), "l4": (64, 128, 1200, 256, 25, 4, 11), "l5": (64, 256, 300, 512, 25, 4, 11)}
b_, cin, Lx, cout, ks, s_, p_ = CASES[os.environ.get("CASE", "tcn3b")]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def stamps(fn):
    L.m2d_debug_stamps_reset()
    fn(); torch.cuda.synchronize()
    L.m2d_debug_stamps(buf, 8192)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:, :4].astype(np.float64) * 0.01
    s = s[(s[:, 0] > 0) & (s[:, 3] > 0)]
    if len(s) == 0: return "no stamps"
    t0 = s[:, 0].min()
    q = lambda x: "%.1f/%.1f/%.1f" % (np.min(x), np.median(x), np.max(x))
    return "span %6.1f, %4d wg | start %s | pro %s | loop %s | epi %s" % (
        s[:, 3].max() - t0, len(s), q(s[:, 0] - t0), q(s[:, 1] - s[:, 0]), q(s[:, 2] - s[:, 1]), q(s[:, 3] - s[:, 2]))


x = torch.randn(b_, cin, Lx, device=dev)
w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
bias = torch.randn(cout, device=dev)
Lout = (Lx + 2 * p_ - ks) // s_ + 1
dy = torch.randn(b_, cout, Lout, device=dev)
gf = 2.0 * b_ * Lout * cout * cin * ks / 1e9
PLANS = [None, (128, 1), (128, 2), (128, 3), (128, 4), (128, 7), (128, 8), (64, 1), (64, 2), (64, 3), (64, 4), (64, 7), (32, 1), (32, 2)]
passes = [("fwd ", lambda: K.conv1d_fwd(x, w, bias, s_, p_, 1, 0.0)),
          ("bwdD", lambda: K.conv1d_bwd_data(dy, w, Lx, s_, p_)),
          ("bwdW", lambda: K.conv1d_bwd_weight(x, dy, ks, s_, p_))]
with K.weight_cache():
    for pname, fn in passes:
        for pl in PLANS:
            if pl is None: os.environ.pop("M2D_PLAN", None)
            else: os.environ["M2D_PLAN"] = "%d,%d" % pl
            try:
                us = timeit(fn)
                print("%s plan %-9s %7.1f us %6.1f TF | %s" % (pname, pl, us, gf / us * 1e3, stamps(fn)), flush=True)
            except Exception as e:
                print(pname, pl, "failed:", str(e)[:80])
