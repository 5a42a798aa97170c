"""Dev tool: which hardware queue did each of the engine's streams land on? (rocprofv3 kernel trace of tools/steady.py)
marker kernels: thin_fwd = main, copyBuffer = copy stream, rowsums_stage2 = generator stream, 4-workgroup persistent
GRU = the generator's noise stream, pose_pack3 = the critic's pose-branch stream."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[-1]
m = collections.defaultdict(collections.Counter)
rows = list(csv.DictReader(open(f)))
for r in rows[len(rows) // 2:]:
    n = r['Kernel_Name']; q = r['Queue_Id']
    if 'thin_fwd' in n: m['main'][q] += 1
    elif 'copyBuffer' in n: m['copy/other'][q] += 1
    elif 'rowsums_stage2' in n: m['gen'][q] += 1
    elif 'gru_persist_fwd' in n and int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) <= 8: m['noise'][q] += 1
    elif 'pose_pack3' in n: m['stick'][q] += 1
print({k: dict(v) for k, v in m.items()})
