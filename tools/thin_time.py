"""Dev tool: event-timed durations of the thin (Cin = 1) conv kernels at the phase-3 size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B = 64
x = torch.randn(B, 1, 76800, device=dev); w = torch.randn(32, 1, 25, device=dev) * 0.2; b = torch.randn(32, device=dev)
dy = torch.randn(B, 32, 19200, device=dev); mask = torch.randn(B, 32, 19200, device=dev)
for _ in range(3):
    K.conv1d_fwd(x, w, b, 4, 11, act=1); K.conv1d_bwd_data(dy, w, 76800, 4, 11, dy_mask=mask); K.conv1d_bwd_weight(x, dy, 25, 4, 11, dy_mask=mask)
torch.cuda.synchronize()
K.prof_begin()
for _ in range(5):
    K.conv1d_fwd(x, w, b, 4, 11, act=1)
    K.conv1d_fwd(x, w, None, 4, 11, out_mask=mask, out_mask_slope=0.0)
    K.conv1d_bwd_data(dy, w, 76800, 4, 11, dy_mask=mask)
    K.conv1d_bwd_weight(x, dy, 25, 4, 11, dy_mask=mask)
    K.conv1d_bwd_weight(x, dy, 25, 4, 11, with_bias=True)
    K.conv1d_bwd_data(dy, w, 76800, 4, 11)
torch.cuda.synchronize()
rows = K.prof_dump(); K.prof_end()
import collections
agg = collections.OrderedDict()
for fam, tag, d0, d1, d2, ms, fl, by in rows:
    a = agg.setdefault((tag, int(by / 1e6)), []); a.append(ms)
for k, v in agg.items(): print("%-24s %4d MB n=%d min %.1f us  median %.1f us" % (k[0], k[1], len(v), 1e3 * min(v), 1e3 * sorted(v)[len(v) // 2]))
