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
# the engine family = what bench.py's `roofline` prices (every dense-contraction kernel: the gather GEMM kernels, the
# tap-vectorised and phase-major conv kernels, the split-K reduction, the TemporalBlock kernels of csrc/tcn.hip)
ENGINE = ("m2d_gemm", "m2d_conv_k4", "m2d_splitk_reduce", "m2d_tcn_")
fam_f = sum(v for k, v in fetch.items() if k.startswith(ENGINE))
fam_w = sum(v for k, v in write.items() if k.startswith(ENGINE))
fam_n = sum(v for k, v in nf.items() if k.startswith(ENGINE) and "reduce" not in k)   # (a reduction belongs to its GEMM's launch)
fam = {"launches": fam_n, "fetch_MB_per_launch_raw": round(fam_f / max(fam_n, 1) / 1e3, 3),
       "fetch_MB_per_launch_x2": round(2 * fam_f / max(fam_n, 1) / 1e3, 3), "write_MB_per_launch": round(fam_w / max(fam_n, 1) / 1e3, 3),
       "fetch_GB_total_x2": round(2 * fam_f / 1e6, 3), "write_GB_total": round(fam_w / 1e6, 3)}
json.dump({"unit": "MB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE are KB)", "engine_family": fam, "per_kernel": out,
           "hbm_bytes_per_launch_uncorrected": 1e6 * (fam["fetch_MB_per_launch_raw"] + fam["write_MB_per_launch"]),
           "hbm_bytes_per_launch_fetch_x2": 1e6 * (fam["fetch_MB_per_launch_x2"] + fam["write_MB_per_launch"])},
          sys.stdout, indent=1)
