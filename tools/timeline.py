"""Dev tool: from a rocprofv3 kernel trace of bench.py, split steady-state wall time into
  big      - at least one kernel with >= 512 workgroups is running (the chip can be full),
  small    - only kernels with fewer workgroups are running (latency- / launch-bound stretch),
  idle     - nothing is running,
and attribute the `small` time to the kernels that were running alone.
    python tools/timeline.py <rocprof output dir>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = []
for r in csv.DictReader(open(f)):
    wg = 1
    for ax in "XYZ":
        g, w = int(r.get("Grid_Size_" + ax, 1) or 1), int(r.get("Workgroup_Size_" + ax, 1) or 1)
        wg *= max(1, (g + w - 1) // w)
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], wg, r.get('Queue_Id', '0')))
rows.sort()
n = len(rows)
rows = rows[n // 2:]
ev = []
for i, (s, e, name, wg, q) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
running = set()
t_prev = ev[0][0]
tot = collections.Counter(); small_by = collections.Counter()
for t, kind, i in ev:
    dt = t - t_prev
    if dt > 0:
        if not running:
            tot["idle"] += dt
        elif any(rows[j][3] >= 512 for j in running):
            tot["big"] += dt
        else:
            tot["small"] += dt
            key = " + ".join(sorted(set(rows[j][2].split("(")[0][-40:] for j in running)))
            small_by[key] += dt
    t_prev = t
    if kind: running.add(i)
    else: running.discard(i)
span = sum(tot.values())
print("span %.2f ms: big %.1f%%  small %.1f%%  idle %.1f%%  (%d kernels)" % (span / 1e6, 100 * tot["big"] / span, 100 * tot["small"] / span, 100 * tot["idle"] / span, len(rows)))
for k, v in small_by.most_common(22):
    print("  %7.3f ms  %s" % (v / 1e6, k))
