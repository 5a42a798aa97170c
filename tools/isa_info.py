"""Dev / test tool (no GPU needed): per-kernel register, LDS and scratch usage and the K-loop instruction mix of the
gfx950 code objects hipcc produced for music2dance_amd/csrc/*.hip (read from music2dance_amd/lib/obj/*.o).

    python tools/isa_info.py [gemm_engine] [--loops]      # table of kernels; --loops: the hot loop of each engine kernel

Used by tests/test_isa_pins.py: the engine's schedule is fragile (unrelated source edits moved the plain
GEMM by 7 %), so the build pins what the hot instantiations must keep - VGPR budget, no scratch, the MFMA count per
16-deep chunk, no `s_waitcnt vmcnt(0)` between the first and the last MFMA of the loop body."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "music2dance_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def code_object(stem):
    """-> path of the gfx950 code object unbundled from lib/obj/<stem>.o (in a temp dir the caller may delete)"""
    src = os.path.join(OBJ, stem + ".o")
    if not os.path.exists(src):
        raise FileNotFoundError(src + " (run python -m music2dance_amd.build)")
    tmp = tempfile.mkdtemp(prefix="m2d_isa_")
    dst = os.path.join(tmp, stem + ".o")
    shutil.copy(src, dst)
    subprocess.run([_tool("llvm-objdump"), "--offloading", dst], check=True, capture_output=True, cwd=tmp)
    for f in os.listdir(tmp):
        if "gfx950" in f:
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 bundle in " + src)


def demangle(names):
    out = subprocess.run([_tool("llvm-cxxfilt") or "c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines()


def kernel_table(co):
    """-> {demangled name: {vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill, symbol}}"""
    txt = subprocess.run([_tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("symbol"):   # first key of the next kernel's block
            rows.append(cur)
            cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                 "vgpr_spill_count", "sgpr_spill_count"):
            cur[k] = int(v)
        elif k == "symbol":
            cur["symbol"] = v.strip("'\"")
        elif k == "name" and "name" not in cur and v.startswith("_Z"):
            cur["name"] = v
    if cur.get("symbol"):
        rows.append(cur)
    names = demangle([r.get("name", r["symbol"].replace(".kd", "")) for r in rows])
    out = {}
    for r, n in zip(rows, names):
        out[n] = {"vgpr": r.get("vgpr_count", -1), "agpr": r.get("agpr_count", 0), "sgpr": r.get("sgpr_count", -1),
                  "lds": r.get("group_segment_fixed_size", 0), "scratch": r.get("private_segment_fixed_size", 0),
                  "vgpr_spill": r.get("vgpr_spill_count", 0), "sgpr_spill": r.get("sgpr_spill_count", 0),
                  "symbol": r["symbol"].replace(".kd", "")}
    return out


def disassemble(co, symbol):
    """-> [(address, text, branch target address or None)] of one kernel (llvm-objdump prints no labels: an
    instruction's address and a branch's target `<symbol+0xoff>` sit in the trailing comment)"""
    txt = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + symbol, co],
                         check=True, capture_output=True, text=True).stdout
    out, base = [], None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line.strip())
        if m:
            base = int(m.group(1), 16)
            continue
        if "//" not in line or base is None:
            continue
        text, comment = line.split("//", 1)
        ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not ma:
            continue
        tgt = None
        mt = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", comment)
        if mt:
            tgt = base + int(mt.group(1), 16)
        elif re.search(r"<[^>+]*>\s*$", comment) and re.match(r"\s*s_c?branch", text):
            tgt = base
        out.append((int(ma.group(1), 16), text.strip(), tgt))
    return out


def hot_loop(ins):
    """The backward-branch loop that holds the most MFMAs (the shortest one among equals) -> (instruction texts, stats)."""
    addr_to_i = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, t, tgt) in enumerate(ins):
        if tgt is None or not re.match(r"s_c?branch", t) or tgt > a or tgt not in addr_to_i:
            continue
        body = [x[1] for x in ins[addr_to_i[tgt]:i + 1]]
        n = sum(1 for x in body if x.startswith("v_mfma"))
        if n and (best is None or n > best[1] or (n == best[1] and len(body) < len(best[0]))):
            best = (body, n)
    if best is None:
        return [], {}
    body = best[0]
    mf = [k for k, x in enumerate(body) if x.startswith("v_mfma")]
    inner = body[mf[0]:mf[-1] + 1]
    stats = {
        "instructions": len(body), "mfma": len(mf),
        "ds_read": sum(1 for x in body if x.startswith("ds_read")),
        "ds_write": sum(1 for x in body if x.startswith("ds_write")),
        "buffer_load": sum(1 for x in body if x.startswith("buffer_load")),
        "lds_dma": sum(1 for x in body if x.startswith("buffer_load") and " lds" in x),
        "valu": sum(1 for x in body if x.startswith("v_") and not x.startswith("v_mfma")),
        "salu": sum(1 for x in body if x.startswith("s_") and not x.startswith("s_waitcnt") and not x.startswith("s_barrier")),
        "barriers": sum(1 for x in body if x.startswith("s_barrier")),
        # a full vector-memory drain BETWEEN the first and the last MFMA serialises staging and multiplying
        "vmcnt0_inside_mfma_span": sum(1 for x in inner if re.match(r"s_waitcnt\b.*vmcnt\(0\)", x)),
        "scratch_ops": sum(1 for x in body if x.startswith("scratch_")),
    }
    return body, stats


def main():
    stems = [a for a in sys.argv[1:] if not a.startswith("-")] or ["gemm_engine"]
    loops = "--loops" in sys.argv
    for stem in stems:
        co = code_object(stem)
        tab = kernel_table(co)
        print("== %s: %d kernels" % (stem, len(tab)))
        print("%-78s %5s %5s %6s %7s %6s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "spill"))
        for n, r in sorted(tab.items()):
            print("%-78s %5d %5d %6d %7d %3d/%-3d" % (n[:78], r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["vgpr_spill"], r["sgpr_spill"]))
            if loops and ("m2d_gemm" in n or "m2d_conv_k4" in n):
                _, st = hot_loop(disassemble(co, r["symbol"]))
                print("      loop:", st)
        shutil.rmtree(os.path.dirname(co), ignore_errors=True)


if __name__ == "__main__":
    main()
