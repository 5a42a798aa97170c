"""Dev tool: per-loop-body summary of a rocprofv3 kernel_stats.csv (bench.py run with --steps S --warmup W)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
bodies = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
m2d = sum(int(r['Calls']) for r in rows if 'm2d' in r['Name'] or 'thin' in r['Name'])
print("sum of kernel time %.2f ms/body, kernels/body %.0f (m2d %.0f, other %.0f)" % (
    tot / 1e6 / bodies, sum(int(r['Calls']) for r in rows) / bodies, m2d / bodies, (sum(int(r['Calls']) for r in rows) - m2d) / bodies))
for r in rows[:top]:
    print("%-64s %6.1f/body %7.3f ms/body %8.1f us" % (r['Name'][:64], int(r['Calls']) / bodies, float(r['TotalDurationNs']) / 1e6 / bodies, float(r['AverageNs']) / 1e3))
