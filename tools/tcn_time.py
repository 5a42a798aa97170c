"""Dev tool: event-timed durations of the pose critic's k7 TemporalBlock launches (forward with two outputs, backward-data
with a masked dy / output mask / skip gradient, the B-row tangent, the weight gradient with its bias column) at the bench
sizes: C3 (B = 64: 192 rows / 64 tangent rows, T = 120), C2 (B = 32), C5 (B = 16, T = 300).

    python tools/tcn_time.py            # the dedicated kernels (csrc/tcn.hip)
    M2D_TCN=0 python tools/tcn_time.py  # the general engine on the same calls
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from music2dance_amd import kernels

K = kernels.impl()
dev = "cuda:0"
REPS = int(os.environ.get("REPS", "20"))


def run(tag, B, T):
    R = 3 * B
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    x, res, mask, mask2 = rnd(R, 128, T), rnd(R, 128, T), rnd(R, 128, T), rnd(R, 128, T)
    dy = rnd(R, 128, T)
    w = rnd(128, 128, 7) * 0.03
    b = rnd(128) * 0.1
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    calls = {
        "fwd relu": lambda: K.conv1d_fwd(x, w, b, 1, 3, 1, out=out),
        "fwd relu + sum_out": lambda: K.conv1d_fwd(x, w, b, 1, 3, 1, residual=res, out=out, sum_out=out2),
        "bwd_data mask/mask": lambda: K.conv1d_bwd_data(dy, w, T, 1, 3, dy_mask=mask, out_mask=mask2, out=out),
        "bwd_data res": lambda: K.conv1d_bwd_data(dy, w, T, 1, 3, residual=res, out=out),
        "tangent (B rows)": lambda: K.conv1d_fwd(x[:B], w, None, 1, 3, 0, out_mask=mask[B:2 * B], out=out[:B]),
        "tangent + res": lambda: K.conv1d_fwd(x[:B], w, None, 1, 3, 0, residual=res[:B], out_mask=mask[B:2 * B], out=out[:B]),
        "bwd_weight + bias": lambda: K.conv1d_bwd_weight(x, dy, 7, 1, 3, with_bias=True, bias_from_sample=B),
        "bwd_weight masked": lambda: K.conv1d_bwd_weight(x, dy, 7, 1, 3, dy_mask=mask, with_bias=True, bias_from_sample=B),
    }
    with K.weight_cache():
        for f in calls.values():
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        for name, f in calls.items():
            K.prof_begin()
            for _ in range(REPS):
                f()
            torch.cuda.synchronize()
            rows = [r for r in K.prof_dump() if r[0] == 0]
            K.prof_end()
            per = len(rows) // REPS
            ms = sorted(sum(r[5] for r in rows[i * per:(i + 1) * per]) for i in range(REPS))
            fl = sum(r[6] for r in rows[:per])
            med = ms[len(ms) // 2]
            print("%-4s %-22s %d launch(es)  min %6.1f us  median %6.1f us  %6.1f TFLOP/s" %
                  (tag, name, per, 1e3 * ms[0], 1e3 * med, fl / (med * 1e-3) / 1e12), flush=True)


print("M2D_TCN =", os.environ.get("M2D_TCN", "1"), " M2D_TCN_NT =", os.environ.get("M2D_TCN_NT", "-"))
for cfg in (os.environ.get("CFGS", "c3,c2,c5")).split(","):
    B, T = {"c3": (64, 120), "c2": (32, 120), "c5": (16, 300)}[cfg]
    run(cfg, B, T)
