"""Dev tool: how much of a loop body runs with the chip under-filled?  From a rocprofv3 kernel trace of tools/steady.py
(streams overlapped), sweep the launch intervals and split the wall time by the number of workgroups the running kernels
asked for: an interval is "thin" when all running kernels together have fewer than THIN workgroups (default 512 = two
per CU). Prints the thin time per body by the set of kernels that were running, i.e. the launches whose latency is NOT
hidden by another stream.      python tools/fill_timeline.py <trace dir> <bodies> [thin]"""
import collections
import csv
import glob
import gzip
import sys

f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv*'))[-1]
bodies = int(sys.argv[2])
thin = int(sys.argv[3]) if len(sys.argv) > 3 else 512
rows = []
for r in csv.DictReader(gzip.open(f, 'rt') if f.endswith('.gz') else open(f)):
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    w = max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], g // w))
rows.sort()
# steady part: tools/steady.py settles the garbage collector (tens of ms with an empty queue) between its warm-up and its
# timed bodies: start behind the last hole of more than 15 ms
end = rows[0][1]
hole, cut = 0, 0
for i, r in enumerate(rows):
    if r[0] - end > 15e6:
        hole, cut = r[0] - end, i
    end = max(end, r[1])
rows = rows[cut:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for i, (s, e, n, wg) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
running = set()
by_set = collections.Counter()
by_fill = collections.Counter()
last = ev[0][0]
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        tot = sum(rows[j][3] for j in running)
        key = 'idle' if not running else ('thin' if tot < thin else 'full')
        by_fill[key] += dt
        if key == 'thin':
            names = tuple(sorted(set(rows[j][2].split('(')[0][-40:] + ':%d' % rows[j][3] for j in running)))
            by_set[names] += dt
    last = t
    if kind:
        running.add(i)
    else:
        running.discard(i)
span = (t1 - t0) / 1e6
nb = bodies
print("span %.2f ms over %d bodies: full %.2f  thin(<%d WGs) %.2f  idle %.2f ms" % (
    span, nb, by_fill['full'] / 1e6, thin, by_fill['thin'] / 1e6, by_fill['idle'] / 1e6))
print("per body: full %.3f  thin %.3f  idle %.3f ms" % (by_fill['full'] / 1e6 / nb, by_fill['thin'] / 1e6 / nb, by_fill['idle'] / 1e6 / nb))
for names, v in by_set.most_common(40):
    print("%8.3f ms/body  %s" % (v / 1e6 / nb, ' + '.join(names)))
