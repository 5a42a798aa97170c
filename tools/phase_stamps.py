"""Dev tool: where does a workgroup of the GEMM engine spend its time? Needs a -DM2D_STAMP build of gemm_engine.hip
(M2D_LIB=<that .so>): every workgroup stamps s_memrealtime (100 MHz) at kernel entry, loop entry, loop exit and after
the epilogue's stores; this prints, per layer and pass of the phase-3 sizes, the kernel span and the per-workgroup
prologue / loop / epilogue times and start offsets (second dispatch waves show up as late starts)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music2dance_amd import kernels, _lib

K = kernels.impl()
L = _lib.lib()
if not hasattr(L, "m2d_debug_stamps"):
    raise SystemExit("not an M2D_STAMP build: set M2D_LIB")
dev = "cuda:0"
B = int(os.environ.get("B", 64))
N = B * 120
CASES = [
    ("temporal.k7 3B", 3 * B, 128, 120, 128, 7, 1, 3),
    ("audio_d.l2", B, 32, 19200, 64, 25, 4, 11),
    ("audio_d.l3", B, 64, 4800, 128, 25, 4, 11),
    ("audio_d.l4", B, 128, 1200, 256, 25, 4, 11),
    ("audio_d.l5", B, 256, 300, 512, 25, 4, 11),
    ("enc.c1", N, 32, 64, 64, 4, 2, 1),
    ("enc.c3", N, 128, 16, 256, 4, 2, 1),
    ("enc.c5", N, 512, 4, 1024, 4, 2, 1),
]
buf = (ctypes.c_ulonglong * (8192 * 8))()


def stamps(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.m2d_debug_stamps_reset()
    fn()
    torch.cuda.synchronize()
    L.m2d_debug_stamps(buf, 8192)
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:, :4].astype(np.float64) * 0.01  # us
    ok = (s[:, 0] > 0) & (s[:, 3] > 0)
    s = s[ok]
    if len(s) == 0:
        return "no stamps"
    t0 = s[:, 0].min()
    q = lambda x: "%.1f/%.1f/%.1f" % (np.min(x), np.median(x), np.max(x))
    late = float(np.mean(s[:, 0] - t0 > 5.0))
    return ("span %7.1f us, %4d wg stamped | start %s (late %.2f) | prologue %s | loop %s | epilogue %s" %
            (s[:, 3].max() - t0, len(s), q(s[:, 0] - t0), late, q(s[:, 1] - s[:, 0]), q(s[:, 2] - s[:, 1]), q(s[:, 3] - s[:, 2])))


a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
for mode in (0, 1, 2):
    print("gemm mode %d 4096^3: %s" % (mode, stamps(lambda: K.gemm(mode, a, b))))
for name, b_, cin, Lx, cout, ks, s_, p_ in CASES:
    x = torch.randn(b_, cin, Lx, device=dev)
    w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev)
    Lout = (Lx + 2 * p_ - ks) // s_ + 1
    dy = torch.randn(b_, cout, Lout, device=dev)
    with K.weight_cache():
        K.conv1d_fwd(x, w, bias, s_, p_, act=1)
        print("%-16s fwd  %s" % (name, stamps(lambda: K.conv1d_fwd(x, w, bias, s_, p_, act=1))))
        print("%-16s bwdD %s" % (name, stamps(lambda: K.conv1d_bwd_data(dy, w, Lx, s_, p_))))
        print("%-16s bwdW %s" % (name, stamps(lambda: K.conv1d_bwd_weight(x, dy, ks, s_, p_))))
