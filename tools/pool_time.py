"""Dev tool: MaxPool / Upsample passes at the U-Net's shapes (rows = 4800 * 128), GB/s per pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
K.prof_begin()
for L in (100, 50, 25, 200):
    x = torch.randn(4800, 128, L, device=dev)
    for _ in range(4):
        y = K.upsample2_fwd(x)
        if L % 2 == 0:
            p = K.maxpool2_fwd(x)
    dy = torch.randn_like(y)
    for _ in range(3):
        K.upsample2_bwd(dy)
    del y, dy
torch.cuda.synchronize()
rows = K.prof_dump(); K.prof_end()
import collections
agg = collections.OrderedDict()
i = 0
for fam, tag, d0, d1, d2, ms, fl, by in rows:
    agg.setdefault((tag, int(by)), []).append((ms, by))
for k, v in agg.items():
    ms = min(m for m, _ in v)
    print("%-16s %8.1f MB  min %7.1f us  %6.0f GB/s" % (k[0], k[1] / 1e6, 1e3 * ms, v[0][1] / ms / 1e6))
