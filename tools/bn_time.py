"""Dev tool: event-timed BatchNorm statistics / backward reductions at the decoder's and encoder's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
shapes = [(7680, 256), (7680, 256, 8), (7680, 32, 64), (3840, 256), (4800, 256)]
xs = [torch.randn(*s, device=dev) for s in shapes]
for x in xs: K.bn_stats(x)
torch.cuda.synchronize()
K.prof_begin()
for _ in range(5):
    for x in xs: K.bn_stats(x)
torch.cuda.synchronize()
rows = K.prof_dump(); K.prof_end()
import collections
agg = collections.OrderedDict()
for fam, tag, d0, d1, d2, ms, fl, by in rows:
    agg.setdefault((tag, d0, d1, d2), []).append(ms)
for k, v in agg.items(): print("%-34s n=%d min %.1f us  median %.1f us" % (k, len(v), 1e3 * min(v), 1e3 * sorted(v)[len(v) // 2]))
