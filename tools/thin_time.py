"""Dev tool: the single-channel k25 / stride-4 layer (audio critic l1, WaveGAN l1) at the step's sizes: forward plain /
masked / with statistics, backward-data, backward-weight - us per launch and TB/s of algorithmic bytes (M2D_LIB per arm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B, L = int(os.environ.get("B", 64)), int(os.environ.get("L", 76800))
x = torch.randn(B, 1, L, device=dev)
w = torch.randn(32, 1, 25, device=dev) * 0.2
b = torch.randn(32, device=dev) * 0.1
y = K.conv1d_fwd(x, w, b, 4, 11, act=2, slope=0.2)
dy = torch.randn_like(y)
MB = y.numel() * 4 / 1e6
forms = [("fwd leaky", lambda: K.conv1d_fwd(x, w, b, 4, 11, act=2, slope=0.2), MB + x.numel() * 4 / 1e6),
         ("fwd masked (tangent)", lambda: K.conv1d_fwd(x, w, None, 4, 11, out_mask=y, out_mask_slope=0.2), 2 * MB + x.numel() * 4 / 1e6),
         ("fwd + statistics", lambda: K.conv1d_fwd(x, w, b, 4, 11, with_stats=True), MB + x.numel() * 4 / 1e6),
         ("bwd_data masked", lambda: K.conv1d_bwd_data(dy, w, L, 4, 11, dy_mask=y, dy_mask_slope=0.2), 2 * MB + x.numel() * 4 / 1e6),
         ("bwd_weight masked", lambda: K.conv1d_bwd_weight(x, dy, 25, 4, 11, dy_mask=y, dy_mask_slope=0.2, with_bias=True), 2 * MB + x.numel() * 4 / 1e6),
         ("bwd_weight", lambda: K.conv1d_bwd_weight(x, dy, 25, 4, 11, with_bias=True), MB + x.numel() * 4 / 1e6)]
for name, f, mb in forms:
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = min(ts)
    print("%-24s %7.1f us  %6.2f TB/s (%.0f MB)" % (name, t, mb / t, mb), flush=True)
