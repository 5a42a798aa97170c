"""Dev tool: what plain fill / copy / read passes reach at the sizes of the step's memory-bound launches (the 157 MB of
bench.py's hbm_probe fits the 256 MiB Infinity Cache; a 430 MB upsample or a 650 MB BatchNorm pass does not).
torch's fill_ / copy_ / sum kernels (16-byte accesses, grid-stride): TB/s over 20 back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
print("%8s %10s %10s %10s %12s" % ("MB", "fill", "copy", "read(sum)", "write 2 : read 1"))
for mb in (40, 80, 157, 240, 320, 430, 650, 980, 1500):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    def timed(f, bytes_):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        return bytes_ * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    fill = timed(lambda: a.fill_(1.0), n * 4)
    h = n // 2
    copy = timed(lambda: b[:h].copy_(a[:h]), n * 4)          # mb total traffic: half read, half written
    read = timed(lambda: a.sum(), n * 4)
    t = n // 3
    up = timed(lambda: torch.cat((a[:t], a[:t]), out=b[:2 * t]), n * 4)   # read a third, write two thirds (an upsample's mix)
    print("%8d %10.2f %10.2f %10.2f %12.2f" % (mb, fill, copy, read, up), flush=True)
    del a, b
