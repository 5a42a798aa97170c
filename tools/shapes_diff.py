"""Dev tool: diff two per-shape tables written by `bench.py --dump-shapes` (same box, two settings): ms per step by shape."""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[(r["family"], r["tag"], r["d0"], r["d1"], r["d2"])] = (float(r["ms_per_step"]), float(r["launches_per_step"]), float(r["us_per_launch"]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in sorted(set(a) | set(b)):
    ma, mb = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    rows.append((mb[0] - ma[0], k, ma, mb))
rows.sort()
tot = sum(r[0] for r in rows)
print("total ms/step: %.3f -> %.3f (%+.3f)" % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values()), tot))
for d, k, ma, mb in rows:
    if abs(d) >= 0.004:
        print("%+7.3f ms  %-10s %-24s %8s %8s %8s  %6.1f us x%.2f -> %6.1f us x%.2f" % (d, k[0], k[1], k[2], k[3], k[4], ma[2], ma[1], mb[2], mb[1]))
