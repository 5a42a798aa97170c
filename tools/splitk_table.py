"""Dev tool: from a rocprofv3 kernel trace, list the split-K reductions of the steady state with the
split factor (grid.z of the GEMM launch before them on the same queue) and their durations."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]
last = {}
agg = collections.defaultdict(list)
for r in rows:
    q = r.get('Queue_Id', '0')
    name = r['Kernel_Name']
    if 'm2d_gemm_kernel' in name:
        last[q] = r
    elif 'm2d_splitk_reduce' in name and q in last:
        g = last[q]
        wg = int(g['Workgroup_Size_X'])
        key = (g['Kernel_Name'].split('(')[0][-34:], int(g['Grid_Size_X']) // wg, int(g['Grid_Size_Y']), int(g['Grid_Size_Z']),
               int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("split-K reductions: %d launches, %.1f us total in the window" % (sum(len(v) for v in agg.values()), tot))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print("%-36s tiles %4d x %3d splits %4d reduce blocks %5d  n=%3d avg %6.1f us  sum %7.1f" % (k + (len(v), sum(v) / len(v), sum(v))))
