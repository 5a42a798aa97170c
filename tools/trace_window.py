"""Dev tool: print every kernel of a rocprofv3 kernel trace inside a time window (ms from the steady window's start,
as tools/queue_busy.py prints them):   python tools/trace_window.py <dir> <bodies> <t0_ms> <t1_ms>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[-1]
bodies = int(sys.argv[2]); a = float(sys.argv[3]) * 1e6; b = float(sys.argv[4]) * 1e6
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'),
         int(r.get('Grid_Size_X', 0) or 0) // max(1, int(r.get('Workgroup_Size_X', 1) or 1)) * max(1, int(r.get('Grid_Size_Y', 1) or 1)) * max(1, int(r.get('Grid_Size_Z', 1) or 1)))
        for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[int(len(rows) * 16 / (16 + bodies)):]
t0 = rows[0][0]
for s, e, n, q, wg in rows:
    if e - t0 >= a and s - t0 <= b:
        print("q%-2s %9.3f -> %9.3f  (%7.1f us) wg %6d  %s" % (q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, wg, n.split("(")[0][-60:]))
