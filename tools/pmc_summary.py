"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the
same bench command). Counter values are KB per dispatch. gfx950: FETCH_SIZE tallies 128-B requests at
64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM section): reported raw and x2."""
import collections, csv, glob, json, sys


def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        k = name.split('(')[0].replace('void ', '')
        if 'm2d_gemm_kernel' in name:
            k = 'm2d_gemm_kernel'
        elif 'at::native' in name or 'rocclr' in name:
            k = 'aten/other'
        tot[k] += float(r['Counter_Value']); n[k] += 1
    return tot, n


fetch, nf = load(sys.argv[1])
write, nw = load(sys.argv[2])
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    out[k] = {"launches": nf[k], "fetch_MB_per_launch_raw": round(fetch[k] / nf[k] / 1e3, 3),
              "fetch_MB_per_launch_x2": round(2 * fetch[k] / nf[k] / 1e3, 3),
              "write_MB_per_launch": round(write.get(k, 0.0) / max(nw.get(k, 1), 1) / 1e3, 3)}
g = out.get('m2d_gemm_kernel', {})
json.dump({"unit": "MB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE are KB)", "per_kernel": out,
           "hbm_bytes_per_launch_uncorrected": 1e6 * (g.get("fetch_MB_per_launch_raw", 0) + g.get("write_MB_per_launch", 0)),
           "hbm_bytes_per_launch_fetch_x2": 1e6 * (g.get("fetch_MB_per_launch_x2", 0) + g.get("write_MB_per_launch", 0))},
          sys.stdout, indent=1)
