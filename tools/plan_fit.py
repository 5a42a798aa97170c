"""Dev tool (CPU): fit the rounds model of csrc/gemm_engine.hip (rounds_cost) on a dump of tools/plan_sweep.py and report,
per swept launch, the plan the fitted model would pick against the fastest measured one.

    python tools/plan_fit.py profiles/r05d_plan_dump.jsonl.gz bwdW      # K-streaming weight gradients (a_kfast cases)
    python tools/plan_fit.py profiles/r05d_plan_dump.jsonl.gz bwdD      # strided backward-data

Model: the workgroups of a launch run in rounds of 256 * R (R = 4 / 7 / 8 resident per CU for 128- / 64- / 32-row tiles);
a round whose CUs hold n workgroups each costs  chunks * max(f, a * n + c) + P  microseconds; split plans add
s0 (x 0.3 when the fix-up runs inside the kernel, <= 16 splits) + s1 per MB of slabs. The printed parameter vector is
[a128 a64 a32  c128 c64 c32  f128 f64 f32  P s0 s1] - the constants kRoundsBwdW / kRoundsBwdD."""
import collections
import gzip
import json
import math
import sys

import numpy as np
from scipy.optimize import least_squares

path, KIND = sys.argv[1], sys.argv[2]
R = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 7, 8]
rows = [json.loads(l) for l in (gzip.open(path, "rt") if path.endswith(".gz") else open(path))]
cases = collections.OrderedDict()
for r in rows:
    if r["kind"] != KIND:
        continue
    if KIND == "fwd" and (".l2" in r["name"] or ".l3" in r["name"]):
        continue  # the tap-vectorised forward has its own launcher (no plan record)
    if KIND == "bwdW" and r["name"] in ("enc.c3", "enc.c4", "enc.c5", "enc.c6"):
        continue  # short outputs: K = (position, sample), another kernel family (plan kind 0)
    c = cases.setdefault(r["name"], dict(plans={}))
    M, N, nch, ph, asplit, pen, bm, sp = r["launch"][:8]
    c["inp"] = (M, N, nch, ph)
    if r["forced"] is None:
        c["auto"] = (bm, sp)
    c["plans"][(bm, sp)] = min(c["plans"].get((bm, sp), 1e9), r["us"])
BI = {128: 0, 64: 1, 32: 2}


def model(x, M, N, nch, ph, bm, sp):
    b = BI[bm]
    a, c, f, P, s0, s1 = x[b], x[3 + b], x[6 + b], x[9], x[10], x[11]
    W = -(-M // bm) * -(-N // 128) * sp * ph
    cps = -(-nch // sp)
    step = lambda n: max(f, a * n + c)
    full = W // (256 * R[b])
    rem = W - full * 256 * R[b]
    tot = full * (cps * step(R[b]) + P)
    if rem > 0:
        tot += cps * step(math.ceil(rem / 256.0)) + P
    if sp > 1:
        tot += s0 * (1.0 if sp > 16 else 0.3) + s1 * sp * M * N * 4 / 1e6
    return 6.0 + tot


pts = [(c["inp"], k, us) for c in cases.values() for k, us in c["plans"].items()]
resid = lambda x: [math.log(model(x, *inp, *k) / us) for inp, k, us in pts]
x0 = [0.98, 0.57, 0.33, 0.5, 0.1, 0.05, 1.46, 1.3, 1.2, 8.0, 8.0, 0.5]
lb = [0.5, 0.3, 0.15, 0, 0, 0, 0.5, 0.5, 0.5, 0, 0, 0]
ub = [1.5, 1.0, 0.8, 1.5, 1.5, 1.5, 2.5, 2.5, 2.5, 40, 40, 5]
x = least_squares(resid, x0, bounds=(lb, ub), loss="soft_l1", f_scale=0.05).x
res = np.array(resid(x))
print("%d points, R = %s" % (len(pts), R))
print("params", np.round(x, 3).tolist())
print("rms log error %.3f  max %.3f" % (np.sqrt((res ** 2).mean()), np.abs(res).max()))
total = 0.0
for name, c in cases.items():
    cost = lambda k: model(x, *c["inp"], *k)
    pick = min(c["plans"], key=cost)
    best = min(c["plans"], key=c["plans"].get)
    loss = c["plans"][pick] / c["plans"][best] - 1
    total += loss
    print("%-14s pick %-10s %7.1f us (model %7.1f) | best %-10s %7.1f | round-4 model %-10s %7.1f | loss %.1f%%" % (
        name, pick, c["plans"][pick], cost(pick), best, c["plans"][best], c.get("auto"), c["plans"].get(c.get("auto"), float("nan")), 100 * loss))
print("sum of losses %.3f" % total)
