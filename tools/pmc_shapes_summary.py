"""Summarise the PMC passes of tools/pmc_shapes.sh: per dispatch of an m2d kernel, duration, MFMA
busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs vs GRBM_GUI_ACTIVE / 8 XCDs), sustained clock,
instructions per MFMA, LDS bank conflicts, and HBM bytes (FETCH_SIZE raw and x2, WRITE_SIZE)."""
import collections, csv, glob, json, sys
root = sys.argv[1]
rows = collections.OrderedDict()
dur = {}
for g in sorted(glob.glob(root + '/g*')):
    f = glob.glob(g + '/*/*counter_collection.csv')
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        k = int(r['Dispatch_Id'])
        e = rows.setdefault(k, {"kernel": r['Kernel_Name'].split('(')[0].replace('void ', '')[:70], "grid": int(r['Grid_Size']),
                                "wg": int(r.get('Workgroup_Size', 256) or 256)})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    t = glob.glob(g + '/*/*kernel_trace.csv')
    if t and not dur:
        for r in csv.DictReader(open(t[0])):
            dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
out = []
for k, v in rows.items():
    if not any(t in v["kernel"] for t in ('m2d', 'k_dl', 'k_guide')):  # (k_*: the kernels of tools/probes/gemm_ceiling.hip)
        continue
    e = {"dispatch": k, "kernel": v["kernel"], "workgroups": v["grid"] // max(v.get("wg", 256), 1), "us": round(dur.get(k, 0.0), 1)}
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui > 0:
        e["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / gui, 3)
        if e["us"] > 0:
            e["clock_GHz"] = round(gui / e["us"] / 1e3, 2)
    m = v.get("SQ_INSTS_MFMA", 0.0)
    if m > 0:
        e["per_mfma"] = {n: round(v.get("SQ_INSTS_" + n, 0.0) / m, 2) for n in ("VALU", "SALU", "LDS", "VMEM_RD")}
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
        w = v["SQ_WAVE_CYCLES"]
        e["wave_cycles"] = {"wait_any": round(v.get("SQ_WAIT_ANY", 0) / w, 2), "wait_inst": round(v.get("SQ_WAIT_INST_ANY", 0) / w, 2),
                            "active": round(v.get("SQ_ACTIVE_INST_ANY", 0) / w, 2)}
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_bank_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in v:
        e["fetch_MB_raw"] = round(v["FETCH_SIZE"] / 1e3, 2)
        e["fetch_MB_x2"] = round(2 * v["FETCH_SIZE"] / 1e3, 2)
    if "WRITE_SIZE" in v:
        e["write_MB"] = round(v["WRITE_SIZE"] / 1e3, 2)
    out.append(e)
cal = [e for e in out if e["kernel"].startswith("m2d_bn_reduce_kernel") and "fetch_MB_raw" in e]
note = {}
if len(cal) >= 2:
    small = [e for e in cal if e["workgroups"] and e["fetch_MB_raw"]]
    note = {"what": "m2d_bn_reduce_kernel reads its input once: 62.9 MB through 4-B-per-lane loads (L = 2) and 78.6 MB through 16-B loads (L = 4800), inputs evicted from the Infinity Cache before each launch",
            "launches": [{"workgroups": e["workgroups"], "fetch_MB_raw": e["fetch_MB_raw"]} for e in small]}
json.dump({"fetch_size_calibration": note, "dispatches": out}, sys.stdout, indent=1)
