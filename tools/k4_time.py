"""Dev tool: the tap-vectorised stride-4 forward on the audio critic's l2 / l3 shapes (B = 64), us per launch.
M2D_K4_PAIR=0/1 and M2D_LIB select the build / K order (one process per arm)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
K = kernels.impl()
for name, B, cin, L, cout in (("l2", 64, 32, 19200, 64), ("l3", 64, 64, 4800, 128)):
    x = torch.randn(B, cin, L, device="cuda"); w = torch.randn(cout, cin, 25, device="cuda") / math.sqrt(cin * 25)
    b = torch.randn(cout, device="cuda")
    with K.weight_cache():
        for _ in range(5): K.conv1d_fwd(x, w, b, 4, 11, 1, 0.0)
        torch.cuda.synchronize()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): K.conv1d_fwd(x, w, b, 4, 11, 1, 0.0)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        gf = 2.0 * B * (L // 4) * cout * cin * 25 / 1e9
        print("%s pair=%s lib=%s: median %.1f us (min %.1f) = %.1f TFLOP/s" % (name, os.environ.get("M2D_K4_PAIR", "1"), os.path.basename(os.environ.get("M2D_LIB", "default")), ts[3], ts[0], gf / ts[3] * 1e3 / 1e3), flush=True)
