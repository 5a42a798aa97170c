"""Dev tool: the WaveGAN encoder's first conv (3 840 windows of 3 200 samples, 794 outputs each) through the thin
forward kernel: dense input vs window view, with / without the BatchNorm statistics, and a 796-position variant
(16-byte aligned rows) for comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B, T, hop, win = 32, 120, 640, 3200
track = torch.randn(B, (T - 1) * hop + win, device=dev)
dense = track.unfold(-1, win, hop).contiguous().view(B * T, 1, win)
dense8 = torch.randn(B * T, 1, win + 8, device=dev)  # Lout = 796
w = torch.randn(32, 1, 25, device=dev) * 0.2; b = torch.randn(32, device=dev)
cases = {
    "dense": lambda: K.conv1d_fwd(dense, w, b, 4, 0, act=1),
    "dense+stats": lambda: K.conv1d_fwd(dense, w, b, 4, 0, with_stats=True),
    "windows": lambda: K.conv1d_fwd_windows(track, T, hop, win, w, b, 4, 0, act=1),
    "windows+stats": lambda: K.conv1d_fwd_windows(track, T, hop, win, w, b, 4, 0, with_stats=True),
    "dense Lout=796": lambda: K.conv1d_fwd(dense8, w, b, 4, 0, act=1),
}
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print("%-16s %.1f us" % (name, a.elapsed_time(e) * 100))
