"""Dev tool: launches per steady-state loop body from a rocprofv3 kernel trace of tools/steady.py (16 warm-up bodies,
then N traced ones - the window is the last N / (16 + N) of the launches, a whole number of 8+1 cycles when N is a
multiple of 8): m2d kernels vs everything else (ATen / runtime fills and copies), by name.
    python tools/tail_count.py <rocprof output dir> [bodies=16]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[-1]
bodies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = [(int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[int(len(rows) * 16 / (16 + bodies)):]
own = collections.Counter(); other = collections.Counter()
for _, n in rows:
    (own if ('m2d' in n or 'thin_' in n) else other)[n.split('(')[0][-90:]] += 1
print("launches per loop body over %d steady bodies: %.1f m2d + %.1f other" % (bodies, sum(own.values()) / bodies, sum(other.values()) / bodies))
print("non-m2d launches:")
for n, c in other.most_common():
    print("  %6.2f/body  %s" % (c / bodies, n))
print("m2d launches:")
for n, c in own.most_common(40):
    print("  %6.2f/body  %s" % (c / bodies, n))
