"""Dev tool: event-timed channel-sum / BatchNorm passes at the phase-3 sizes."""
import sys, os
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
for (B, C, L) in ((64, 32, 19200), (64, 64, 4800), (64, 128, 1200), (64, 512, 75), (128, 128, 120), (7680, 64, 32), (7680, 256, 8), (7680, 1024, 2), (7680, 256, 1)):
    x = torch.randn(B, C, L, device=dev); m = torch.randn(B, C, L, device=dev)
    mb = x.numel() * 4 / 1e6
    t1 = timeit(lambda: K.channel_sums(x)); t2 = timeit(lambda: K.channel_sums(x, m, 0.0))
    g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    t3 = timeit(lambda: K.bn_fwd(x, g, bt, rm, rv, True, 1e-5, 0.1, act=1))
    print("B%5d C%5d L%6d %6.1f MB | sums %6.1f us %5.2f TB/s | masked %6.1f us %5.2f TB/s | bn_fwd %6.1f us %5.2f TB/s (3 passes)" % (
        B, C, L, mb, t1, mb / t1, t2, 2 * mb / t2, t3, 3 * mb / t3))
