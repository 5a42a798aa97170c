"""Dev tool: event-timed BatchNorm backward (reduce + apply) and statistics at the U-Net / audio shapes with long rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
shapes = [(4800, 128, 200), (4800, 128, 100), (4800, 64, 400), (4800, 32, 800), (4800, 128, 48), (64, 64, 4800)]
K.prof_begin()
for s in shapes:
    x = torch.randn(*s, device=dev); dy = torch.randn(*s, device=dev)
    C = s[1]
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    for _ in range(4):
        y, mean, invstd = K.bn_fwd(x, g, b, rm, rv, True, 1e-5, 0.1, 1)
        K.bn_bwd(dy, x, g, b, mean, invstd, 1)
torch.cuda.synchronize()
rows = K.prof_dump(); K.prof_end()
import collections
agg = collections.OrderedDict()
for fam, tag, d0, d1, d2, ms, fl, by in rows:
    agg.setdefault((tag, d0, d1, d2), []).append((ms, by))
for k, v in agg.items():
    ms = min(m for m, _ in v)
    print("%-40s min %7.1f us  %6.0f GB/s" % (k, 1e3 * ms, v[0][1] / ms / 1e6))
