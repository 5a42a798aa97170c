"""Bench / dev tool: the shader clock INSIDE the step's largest launches (round-5 verdict item 6). Needs the -DM2D_STAMP
build of the library (python -m music2dance_amd.build --stamp; M2D_LIB=music2dance_amd/lib/libm2d_hip_stamp.so): every
workgroup stamps s_memrealtime (100 MHz) and s_memtime (the shader clock's counter) at loop entry and loop exit; the
clock of a launch = median over its workgroups of d memtime / d realtime. Prints ONE JSON line:
  {"shapes": [{"what", "M", "N", "K", "us", "tflops", "clock_GHz"} ...], "clock_GHz_in_step": launch-time-weighted mean}
Shapes: the five largest engine launches of the phase-3 default workload at B = 64 (profiles/*_shapes_c3.csv), the pose
critic's 3B-row k7 launch (csrc/tcn.hip), the plain 4096^3 GEMM on the engine and on the in-library probe kernel."""
import ctypes, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from music2dance_amd import kernels, _lib

K = kernels.impl()
L = _lib.lib()
if not hasattr(L, "m2d_debug_stamps") or not hasattr(L, "m2d_tcn_stamps"):
    raise SystemExit("not an M2D_STAMP build: set M2D_LIB")
dev = "cuda:0"
B = int(os.environ.get("B", 64))
ebuf = (ctypes.c_ulonglong * (8192 * 8))()
tbuf = (ctypes.c_ulonglong * (4096 * 8))()


def measure(fn, tcn=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    (L.m2d_tcn_stamps_reset if tcn else L.m2d_debug_stamps_reset)()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    if tcn:
        L.m2d_tcn_stamps(tbuf, 4096)
        raw = np.frombuffer(tbuf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
    else:
        L.m2d_debug_stamps(ebuf, 8192)
        raw = np.frombuffer(ebuf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
    raw = raw[(raw[:, 1] > 0) & (raw[:, 2] > raw[:, 1])]
    ghz = float(np.median((raw[:, 6] - raw[:, 5]) / (raw[:, 2] - raw[:, 1])) * 0.1) if len(raw) else None
    return 1e3 * e0.elapsed_time(e1), ghz


rows = []


def add(what, M, N, Kd, fn, tcn=False, flops=None):
    us, ghz = measure(fn, tcn)
    fl = flops if flops is not None else 2.0 * M * N * Kd
    rows.append({"what": what, "M": M, "N": N, "K": Kd, "us": round(us, 1), "tflops": round(fl / us / 1e6, 1),
                 "clock_GHz": None if ghz is None else round(ghz, 3)})


def conv_case(name, b_, cin, Lx, cout, ks, s_, p_, which):
    x = torch.randn(b_, cin, Lx, device=dev)
    w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
    bias = torch.randn(cout, device=dev)
    Lout = (Lx + 2 * p_ - ks) // s_ + 1
    dy = torch.randn(b_, cout, Lout, device=dev)
    with K.weight_cache():
        if which == "fwd":
            add(name + " forward", cout, b_ * Lout, cin * ks, lambda: K.conv1d_fwd(x, w, bias, s_, p_, act=1))
        elif which == "bwd_weight":
            add(name + " weight gradient", cout, cin * ks + 1, b_ * Lout, lambda: K.conv1d_bwd_weight(x, dy, ks, s_, p_, with_bias=True))
        else:  # (reported K = the taps that exist: a strided backward-data multiplies ks / stride taps per output position)
            add(name + " backward-data", cin, b_ * Lx, cout * ks // s_, lambda: K.conv1d_bwd_data(dy, w, Lx, s_, p_),
                flops=2.0 * b_ * Lout * cout * cin * ks)


N = B * 120
conv_case("encoder 512->1024 k4 s2", N, 512, 4, 1024, 4, 2, 1, "fwd")
conv_case("audio critic 32->64 k25 s4", 2 * B, 32, 19200, 64, 25, 4, 11, "bwd_weight")
conv_case("audio critic 128->256 k25 s4", B, 128, 1200, 256, 25, 4, 11, "fwd")
conv_case("audio critic 256->512 k25 s4", 2 * B, 256, 300, 512, 25, 4, 11, "bwd_weight")
conv_case("audio critic 256->512 k25 s4", 2 * B, 256, 300, 512, 25, 4, 11, "bwd_data")
# the pose critic's TemporalBlock launch (its own stamp buffer)
x = torch.randn(3 * B, 128, 120, device=dev)
w = torch.randn(128, 128, 7, device=dev) * 0.03
bb = torch.randn(128, device=dev)
with K.weight_cache():
    add("pose critic k7, 3B rows, forward (csrc/tcn.hip)", 128, 3 * B * 120, 896, lambda: K.conv1d_fwd(x, w, bb, 1, 3, 1), tcn=True)
n = 4096
ga, gb = torch.randn(n, n, device=dev), torch.randn(n, n, device=dev)
add("plain 4096^3 GEMM, engine (mode 2)", n, n, n, lambda: K.gemm(2, ga, gb))
weighted = [(r["us"], r["clock_GHz"]) for r in rows[:6] if r["clock_GHz"]]
out = {"shapes": rows, "batch": B,
       "clock_GHz_in_step": round(sum(u * c for u, c in weighted) / sum(u for u, _ in weighted), 3) if weighted else None,
       "note": "shader clock = median over a launch's workgroups of d s_memtime / d s_memrealtime across the K loop (stamped "
               "build); clock_GHz_in_step = launch-time-weighted mean over the step's six launches above"}
print(json.dumps(out))
