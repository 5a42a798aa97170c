"""Dev tool: the sub-pixel backward-data launch of the 32 -> 64 audio layer (k25, stride 4), phase-major rows without the
phantom taps (M2D_SUBPIXEL_TALL=1, the default) against the (ci, r) order (=0); both checked against fp64 autograd."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from music2dance_amd import kernels
    K = kernels.impl()
    dev = "cuda:0"
    torch.manual_seed(0)
    for b, cin, L, cout in ((64, 32, 19200, 64), (128, 32, 19200, 64), (128, 64, 4800, 128), (128, 128, 1200, 256), (128, 256, 300, 512)):
        ks, s, p = 25, 4, 11
        Lout = (L + 2 * p - ks) // s + 1
        w = torch.randn(cout, cin, ks, device=dev) / 28.0
        dy = torch.randn(b, cout, Lout, device=dev)
        mask = torch.randn(b // 2 if b == 128 else b, cin, L, device=dev)
        fn = lambda: K.conv1d_bwd_data(dy, w, L, s, p, out_mask=mask)
        with K.weight_cache():
            out = fn()
            for _ in range(5): fn()
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) * 100
        ref = torch.nn.grad.conv1d_input((b, cin, L), w.double(), dy.double(), stride=s, padding=p)
        mm = mask if mask.shape[0] == b else torch.cat([mask, mask])
        ref = ref * (mm > 0).double()
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        gf = 2.0 * b * Lout * cout * cin * ks / 1e6  # MFLOP: / us = TFLOP/s
        print("Cin %3d rows %3d  %7.1f us  %6.1f TF   max rel err %.2e" % (cin, b, us, gf / us, err))
else:
    for v, mc in (("0", "32"), ("1", "32"), ("1", "256"), ("0", "32"), ("1", "32"), ("1", "256")):
        print("== M2D_SUBPIXEL_TALL=%s M2D_SUBPIXEL_TALL_MAXCIN=%s" % (v, mc), flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, M2D_SUBPIXEL_TALL=v, M2D_SUBPIXEL_TALL_MAXCIN=mc))
