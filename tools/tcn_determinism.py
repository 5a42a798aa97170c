"""Dev tool: run each TemporalBlock kernel form N times on the same inputs, while a second stream keeps the chip busy, and
report how many repeats differ bit-wise from the first.  python tools/tcn_determinism.py [N]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DEV = "cuda:0"
K = kernels.impl()
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
noise_a = torch.randn(2048, 2048, device=DEV)


def busy():
    with torch.cuda.stream(side):
        for _ in range(3):
            noise_a @ noise_a


def repeat(name, fn):
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    bad = 0
    for i in range(N):
        if BUSY and i % 3 == 0:
            busy()
        out = fn()
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
            if bad <= 3:
                for a, b in zip(out, ref):
                    idx = (a != b).nonzero()
                    if idx.numel():
                        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
                        print("    %d elements differ, index box %s .. %s, worst %.3e (largest |ref| %.3e)" %
                              (idx.shape[0], lo, hi, (a - b).abs().max().item(), b.abs().max().item()), flush=True)
    torch.cuda.synchronize()
    print("%-44s %d / %d repeats differ" % (name, bad, N), flush=True)


SHAPES = [(6, 128, 120), (4, 100, 120), (6, 112, 120), (32, 128, 120)]
if os.environ.get("ALL"):
    SHAPES = [(2, 128, 120), (6, 128, 120), (4, 100, 120), (96, 128, 120), (50, 128, 120), (192, 128, 120), (130, 128, 120)]
BUSY = os.environ.get("BUSY", "1") == "1"
for (B, Cin, L) in SHAPES:
    x = (torch.randn(B, Cin, L, generator=g)).to(DEV)
    w = (torch.randn(128, Cin, 7, generator=g) / math.sqrt(Cin * 7)).to(DEV)
    b = (torch.randn(128, generator=g) * 0.1).to(DEV)
    dy = torch.randn(B, 128, L, generator=g).to(DEV)
    mask = torch.randn(B, Cin, L, generator=g).to(DEV)
    tag = "B%d C%d L%d " % (B, Cin, L)
    repeat(tag + "fwd", lambda: (K.conv1d_fwd(x, w, b, 1, 3, act=1),))
    if Cin == 128:
        repeat(tag + "bwd_data", lambda: (K.conv1d_bwd_data(dy, w, L, 1, 3),))
        repeat(tag + "bwd_data masked", lambda: (K.conv1d_bwd_data(dy, w, L, 1, 3, dy_mask=mask, dy_mask_slope=0.2,
                                                                   out_mask=mask, out_mask_slope=0.1, residual=x),))
    if L % 60 == 0:
        repeat(tag + "bwd_weight", lambda: K.conv1d_bwd_weight(x, dy, 7, 1, 3, dy_mask=dy, dy_mask_slope=0.2, with_bias=True))
