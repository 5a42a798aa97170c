"""Dev tool: where does a workgroup of the TemporalBlock kernels (csrc/tcn.hip) spend its time? Needs a -DM2D_STAMP build
(bash tools/build_variant.sh stamp -DM2D_STAMP; M2D_LIB=music2dance_amd/lib_stamp/libm2d_hip.so): every workgroup stamps
s_memrealtime (100 MHz) at entry, loop entry, loop exit and after its last store."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music2dance_amd import kernels, _lib

K = kernels.impl()
L = _lib.lib()
if not hasattr(L, "m2d_tcn_stamps"):
    raise SystemExit("not an M2D_STAMP build: set M2D_LIB")
dev = "cuda:0"
buf = (ctypes.c_ulonglong * (4096 * 8))()


def stamps(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.m2d_tcn_stamps_reset()
    fn()
    torch.cuda.synchronize()
    L.m2d_tcn_stamps(buf, 4096)
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
    raw = raw[(raw[:, 0] > 0) & (raw[:, 3] > 0)]
    s = raw[:, :4] * 0.01  # us
    ghz = np.median((raw[:, 6] - raw[:, 5]) / np.maximum(raw[:, 2] - raw[:, 1], 1.0)) * 0.1  # shader clocks per 10 ns
    if len(s) == 0:
        return "no stamps"
    t0 = s[:, 0].min()
    q = lambda x: "%.1f/%.1f/%.1f" % (np.min(x), np.median(x), np.max(x))
    return ("span %6.1f us, %4d wg | start %s | prologue %s | loop %s | epilogue %s | clock in the loop %.2f GHz" %
            (s[:, 3].max() - t0, len(s), q(s[:, 0] - t0), q(s[:, 1] - s[:, 0]), q(s[:, 2] - s[:, 1]), q(s[:, 3] - s[:, 2]), ghz))


for tag, B, T in (("c3", 64, 120), ("c2", 32, 120), ("c5", 16, 300)):
    R = 3 * B
    x, res, mask, dy = (torch.randn(R, 128, T, device=dev) for _ in range(4))
    w = torch.randn(128, 128, 7, device=dev) * 0.03
    b = torch.randn(128, device=dev) * 0.1
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    with K.weight_cache():
        print(tag, "fwd relu+sum  ", stamps(lambda: K.conv1d_fwd(x, w, b, 1, 3, 1, residual=res, out=out, sum_out=out2)))
        print(tag, "bwd_data mask ", stamps(lambda: K.conv1d_bwd_data(dy, w, T, 1, 3, dy_mask=mask, out_mask=res, out=out)))
        print(tag, "tangent       ", stamps(lambda: K.conv1d_fwd(x[:B], w, None, 1, 3, 0, out_mask=mask[B:2 * B], out=out[:B])))
        print(tag, "bwd_weight    ", stamps(lambda: K.conv1d_bwd_weight(x, dy, 7, 1, 3, with_bias=True, bias_from_sample=B)))
        print(tag, "bwd_weight msk", stamps(lambda: K.conv1d_bwd_weight(x, dy, 7, 1, 3, dy_mask=mask, with_bias=True, bias_from_sample=B)))
