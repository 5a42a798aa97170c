"""Dev tool: from a rocprofv3 kernel trace of tools/steady.py, per HIP stream (queue): busy time, kernels, and the
time during which ONLY that queue had a kernel running - which stream is the critical path of the loop body?
    python tools/queue_busy.py <rocprof output dir> [bodies=16]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[-1]
bodies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'))
        for r in csv.DictReader(open(f))]
rows.sort()
# the steady window: the last `bodies` of (16 warm-up + bodies) bodies by kernel count
n = len(rows)
rows = rows[int(n * 16 / (16 + bodies)):]
span = rows[-1][1] - rows[0][0]
ev = []
for i, (s, e, name, q) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
running = collections.Counter()
busy = collections.Counter(); alone = collections.Counter(); alone_k = collections.Counter()
idle = 0
t_prev = ev[0][0]
cur = set()
for t, kind, i in ev:
    dt = t - t_prev
    if dt > 0:
        qs = [q for q, c in running.items() if c > 0]
        for q in qs: busy[q] += dt
        if len(qs) == 1:
            alone[qs[0]] += dt
            for j in cur: alone_k[(qs[0], rows[j][2].split("(")[0][-44:])] += dt
        if not qs: idle += dt
    t_prev = t
    if kind: running[rows[i][3]] += 1; cur.add(i)
    else: running[rows[i][3]] -= 1; cur.discard(i)
print("steady span %.2f ms = %.3f ms/body, idle %.2f%%" % (span / 1e6, span / 1e6 / bodies, 100.0 * idle / span))
cnt = collections.Counter(r[3] for r in rows)
for q in sorted(busy, key=lambda q: -busy[q]):
    print("queue %-4s kernels/body %6.1f  busy %6.3f ms/body  alone %6.3f ms/body" % (
        q, cnt[q] / bodies, busy[q] / 1e6 / bodies, alone[q] / 1e6 / bodies))
print("kernels running while theirs is the only busy queue:")
for (q, k), v in alone_k.most_common(24):
    print("  q%-3s %7.3f ms/body  %s" % (q, v / 1e6 / bodies, k))
# merged stretches during which one queue ran alone (>= 0.2 ms), in time order: where in the cycle do they sit?
if len(sys.argv) > 3:
    t0 = rows[0][0]
    cur, running = set(), collections.Counter()
    t_prev = ev[0][0]
    stretch = None
    out = []
    for t, kind, i in ev:
        qs = [q for q, c in running.items() if c > 0]
        lone = qs[0] if len(qs) == 1 else None
        if stretch and (lone != stretch[0]):
            if t_prev - stretch[1] >= 0 : out.append((stretch[1], t_prev, stretch[0], stretch[2]))
            stretch = None
        if lone is not None and stretch is None and t > t_prev:
            stretch = (lone, t_prev, set())
        if stretch:
            for j in cur: stretch[2].add(rows[j][2].split("(")[0][-30:])
        t_prev = t
        if kind: running[rows[i][3]] += 1; cur.add(i)
        else: running[rows[i][3]] -= 1; cur.discard(i)
    # merge neighbours of the same queue separated by < 30 us
    merged = []
    for s, e, q, names in out:
        if merged and merged[-1][2] == q and s - merged[-1][1] < 30000:
            merged[-1] = (merged[-1][0], e, q, merged[-1][3] | names)
        else:
            merged.append((s, e, q, set(names)))
    for s, e, q, names in merged:
        if e - s >= 200000 and q != sys.argv[3]:
            print("  t=%8.2f ms  %6.2f ms  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, ", ".join(sorted(names))[:150]))
