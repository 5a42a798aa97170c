"""Dev tool: from a rocprofv3 kernel trace, the kernels of ONE steady-state loop body in start order: start offset,
duration, gap to the previous kernel's end (all streams merged), name. Args: trace dir, a kernel-name substring that
occurs once per body (the body delimiter), optional body index from the end."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
key = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if key in r[2]]
a, b = marks[-back - 1], marks[-back]
body = rows[a:b]
t0 = body[0][0]
end = t0
busy = 0
print("body of %d kernels, %.1f us" % (len(body), (body[-1][1] - t0) / 1e3))
for s, e, n, q in body:
    gap = (s - end) / 1e3
    print("%8.1f  dur %7.1f  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q[-2:], n[:70]))
    if e > end:
        busy += e - max(s, end)
        end = e
print("busy %.1f us of %.1f" % (busy / 1e3, (end - t0) / 1e3))
