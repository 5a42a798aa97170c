"""Build the gfx950 HIP library (libm2d_hip.so) in-tree with hipcc.

`python -m music2dance_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libm2d_hip.so")
SOURCES = ["m2d_runtime.hip", "gemm_engine.hip", "conv1d.hip", "tcn.hip", "probe_gemm.hip", "conv1d_thin.hip", "bn.hip", "gru.hip",
           "pointwise.hip"]
# the sources that carry -DM2D_STAMP instrumentation (the stamped variant reuses every other object)
STAMPED = ["gemm_engine.hip", "tcn.hip"]
STAMP_LIB_PATH = os.path.join(LIB_DIR, "libm2d_hip_stamp.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"] + os.environ.get("M2D_EXTRA_FLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libm2d_hip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, stamp=True):
    """Compile every .hip source for gfx950 and link libm2d_hip.so (and, with `stamp`, libm2d_hip_stamp.so: the same
    library with the per-workgroup clock stamps of the -DM2D_STAMP build). Returns the product library's path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[m2d build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r

    # the stamped variant (tools/step_clock.py, tools/phase_stamps.py: per-workgroup s_memrealtime / s_memtime stamps - what
    # bench.py reads the in-step shader clock from, in a child process): only the instrumented sources are compiled twice
    sobjs = []
    sjobs = []
    if stamp:
        for src in SOURCES:
            if src in STAMPED:
                s = os.path.join(CSRC, src)
                o = os.path.join(obj_dir, src.replace(".hip", ".stamp.o"))
                sobjs.append(o)
                if force or _stale(o, [s] + headers):
                    sjobs.append([hipcc] + FLAGS + ["-DM2D_STAMP", "-c", s, "-o", o])
            else:
                sobjs.append(os.path.join(obj_dir, src.replace(".hip", ".o")))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs + sjobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    if stamp and (force or jobs or sjobs or _stale(STAMP_LIB_PATH, sobjs)):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STAMP_LIB_PATH] + sobjs)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, stamp="--no-stamp" not in sys.argv)
    print(LIB_PATH)
