"""Dev tool: time conv shapes under forced (bm, splits) plans. Needs a -DM2D_TUNING build."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, json
import torch
from music2dance_amd import kernels, _lib
K = kernels.impl()
DUMP = open(os.environ["DUMP"], "a") if os.environ.get("DUMP") else None
def last_plan():
    out = (ctypes.c_int * 9)()
    try:
        _lib.lib().m2d_debug_last_plan(out)
    except AttributeError:
        return None
    return list(out)
dev = "cuda:0"

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

B = 64
N = B * 120
CASES = [("temporal.k7 B32", 32, 128, 120, 128, 7, 1, 3), ("temporal.k7 B64", 64, 128, 120, 128, 7, 1, 3),
         ("temporal.k7 B96", 96, 128, 120, 128, 7, 1, 3), ("temporal.k7 B128", 128, 128, 120, 128, 7, 1, 3),
         ("temporal.k7 B192", 192, 128, 120, 128, 7, 1, 3), ("temporal.k7 C5 48x300", 48, 128, 300, 128, 7, 1, 3),
         ("temporal.k7 C5 16x300", 16, 128, 300, 128, 7, 1, 3),
         ("stick.conv1 B64", 64, 69, 120, 128, 25, 1, 12),
         ("audio_d.l2", B, 32, 19200, 64, 25, 4, 11), ("audio_d.l3", B, 64, 4800, 128, 25, 4, 11),
         ("audio_d.l4", B, 128, 1200, 256, 25, 4, 11), ("audio_d.l5", B, 256, 300, 512, 25, 4, 11),
         ("audio_d2B.l2", 2 * B, 32, 19200, 64, 25, 4, 11), ("audio_d2B.l3", 2 * B, 64, 4800, 128, 25, 4, 11),
         ("audio_d2B.l4", 2 * B, 128, 1200, 256, 25, 4, 11), ("audio_d2B.l5", 2 * B, 256, 300, 512, 25, 4, 11),
         ("audio_dh.l2", B // 2, 32, 19200, 64, 25, 4, 11), ("audio_dh.l3", B // 2, 64, 4800, 128, 25, 4, 11),
         ("audio_dh.l4", B // 2, 128, 1200, 256, 25, 4, 11), ("audio_dh.l5", B // 2, 256, 300, 512, 25, 4, 11),
         ("wavegan.l4", 3840, 128, 43, 256, 25, 4, 0), ("wavegan.l3", 3840, 64, 193, 128, 25, 4, 0),
         ("enc.c1", N, 32, 64, 64, 4, 2, 1), ("enc.c2", N, 64, 32, 128, 4, 2, 1), ("enc.c3", N, 128, 16, 256, 4, 2, 1), ("enc.c4", N, 256, 8, 512, 4, 2, 1),
         ("enc.c5", N, 512, 4, 1024, 4, 2, 1), ("enc.c6", N, 1024, 2, 250, 2, 1, 0)]
PLANS = [None, (128, 1), (64, 1), (32, 1), (128, 2), (64, 2), (32, 2), (128, 4), (64, 4), (32, 4), (128, 8), (64, 8), (128, 16), (64, 16), (128, 32), (64, 32), (128, 64), (64, 64), (128, 3), (64, 3), (128, 6), (64, 6), (128, 12), (64, 12), (128, 24), (64, 24), (128, 48), (64, 48), (128, 96), (64, 96), (128, 128), (64, 128), (64, 192), (64, 256)]
only = os.environ.get("CASE")
for name, b, cin, L, cout, ks, s, p in CASES:
    if only and only not in name: continue
    x = torch.randn(b, cin, L, device=dev); w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev); Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev); gf = 2.0 * b * Lout * cout * cin * ks / 1e9
    print("==", name, "GF %.2f" % gf)
    res = {0: [], 1: [], 2: []}
    with K.weight_cache():
        fns = [lambda: K.conv1d_fwd(x, w, bias, s, p, act=1), lambda: K.conv1d_bwd_data(dy, w, L, s, p),
               lambda: K.conv1d_bwd_weight(x, dy, ks, s, p)]
        os.environ.pop("M2D_PLAN", None)
        for f in fns:            # clocks up before the first timed plan
            for _ in range(10): f()
        for pl in PLANS + [None]:   # (the model's own plan is timed first AND last)
            if pl is None: os.environ.pop("M2D_PLAN", None)
            else: os.environ["M2D_PLAN"] = "%d,%d" % pl
            for i, f in enumerate(fns):
                res[i].append((min(timeit(f, 6), timeit(f, 6)), pl))
                if DUMP:
                    DUMP.write(json.dumps(dict(name=name, kind=("fwd", "bwdD", "bwdW")[i], b=b, cin=cin, L=L, cout=cout, ks=ks, s=s, p=p,
                                               forced=pl, us=1e3 * res[i][-1][0], launch=last_plan())) + "\n")
                    DUMP.flush()
    os.environ.pop("M2D_PLAN", None)
    for i, nm in enumerate(("fwd", "bwdD", "bwdW")):
        auto = min(res[i][0][0], res[i][-1][0])
        best = sorted(r for r in res[i] if r[1] is not None)[:4]
        flag = " <<<" if best[0][0] < 0.96 * auto else ""
        print("  %-5s auto %7.1f us %5.1f TF | best " % (nm, 1e3 * auto, gf / auto) + "  ".join("%s %.1f us %.1f TF" % (pl, 1e3 * t, gf / t) for t, pl in best) + flag, flush=True)
