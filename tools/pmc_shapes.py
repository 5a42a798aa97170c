"""Dev tool: a few representative launches for rocprofv3 --pmc passes (tools/pmc_shapes.sh):
plain GEMM, audio-critic / encoder / TCN conv shapes at B = 64, and two reductions over tensors of
known size that calibrate FETCH_SIZE for 4-B-per-lane and 16-B-per-lane streaming reads."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
dev = "cuda:0"
B = 64
def conv_case(cin, L, cout, ks, s, p, which, b=B):
    x = torch.randn(b, cin, L, device=dev); w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(b, cout, Lout, device=dev)
    with K.weight_cache():
        for _ in range(3):
            if which == 0: K.conv1d_fwd(x, w, None, s, p, act=1)
            elif which == 1: K.conv1d_bwd_data(dy, w, L, s, p)
            else: K.conv1d_bwd_weight(x, dy, ks, s, p)
    torch.cuda.synchronize()
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
for _ in range(3): K.gemm(0, a, b)
torch.cuda.synchronize()
conv_case(32, 19200, 64, 25, 4, 11, 0)     # audio l2 fwd (round 3: tap-vectorised kernel)
conv_case(64, 4800, 128, 25, 4, 11, 0)     # audio l3 fwd (tap-vectorised)
conv_case(128, 1200, 256, 25, 4, 11, 0)    # audio l4 fwd (generic LDS-direct kernel)
conv_case(256, 300, 512, 25, 4, 11, 0)     # audio l5 fwd
conv_case(32, 19200, 64, 25, 4, 11, 2)     # audio l2 bwd_weight
conv_case(256, 300, 512, 25, 4, 11, 1)     # audio l5 bwd_data
conv_case(512, 4, 1024, 4, 2, 1, 0, b=7680)  # encoder c5 fwd
conv_case(128, 120, 128, 7, 1, 3, 0, b=3 * B)   # pose critic k7, 3B rows: forward (csrc/tcn.hip, 96-column tiles)
conv_case(128, 120, 128, 7, 1, 3, 1, b=3 * B)   # ... backward-data (the same kernel, taps flipped)
conv_case(128, 120, 128, 7, 1, 3, 2, b=3 * B)   # ... weight gradient (m2d_tcn_wgrad_kernel + its reduction)
conv_case(128, 120, 128, 7, 1, 3, 0)            # the B-row tangent (32-column tiles)
# FETCH_SIZE calibration: m2d_bn_reduce_kernel reads every byte of its input exactly once.
#   L = 2    -> dword loads (4 B per lane): 7680 x 1024 x 2 floats = 62.9 MB
#   L = 4800 -> 16-B loads:                 64 x 64 x 4800 floats  = 78.6 MB
xs = torch.randn(7680, 1024, 2, device=dev); xl = torch.randn(64, 64, 4800, device=dev)
big = torch.randn(96, 1024, 1024, device=dev)  # 403 MB: pushes the inputs out of the 256 MB Infinity Cache
for _ in range(3):
    big.add_(1.0); torch.cuda.synchronize(); K.channel_sums(xs)
    big.add_(1.0); torch.cuda.synchronize(); K.channel_sums(xl)
torch.cuda.synchronize()
