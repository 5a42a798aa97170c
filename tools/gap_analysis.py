"""Dev tool: from a rocprofv3 kernel trace of bench.py, report busy vs idle time on the GPU
over the last third of the run (steady state) and the biggest idle gaps with their neighbours."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows)
rows = rows[2 * n // 3:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_end = rows[0][0]; gaps = []
prev = None
for s, e, name in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev, name))
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        prev = name
print("span %.2f ms busy %.2f ms idle %.2f ms (%.1f%%) kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0), len(rows)))
import collections
agg = collections.Counter(); cnt = collections.Counter()
for g, a, b in gaps:
    k = (a[:45], b[:45]); agg[k] += g; cnt[k] += 1
for k, v in agg.most_common(25):
    print("%8.3f ms %5d x  %-45s -> %-45s" % (v / 1e6, cnt[k], k[0], k[1]))
