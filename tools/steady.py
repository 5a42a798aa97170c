"""Dev tool: N steady-state phase-3 loop bodies and nothing else (for rocprofv3 --kernel-trace; bench.py adds its
single-stream roofline pass and the CPU baseline to a trace).   python tools/steady.py [bodies] [preset]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
# A/B aid: HIP maps streams onto GPU_MAX_HW_QUEUES (4) hardware queues; dummy streams created (and used) first shift
# which of the engine's streams end up sharing a queue
_dummies = []
for _i in range(int(os.environ.get("STREAM_SHIFT", "0"))):
    _s = torch.cuda.Stream(dev)
    with torch.cuda.stream(_s):
        torch.zeros(8, device=dev)
    _dummies.append(_s)
torch.cuda.synchronize()
gen, critic = bench.build_models(dev)
if os.environ.get("NOISE_OVERLAP") == "0":  # A/B: the generator's noise GRU in line instead of on its own side stream
    type(gen).overlap_noise_gru = False
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices, ready = synthetic_phase3_batch(64, 120, dev, seed=1, with_event=True)
for _ in range(16): eng.train_step(real, audio, slices, inputs_ready=ready)
from music2dance_amd import runner
runner.settle_garbage_collector()  # (a generation-2 collection inside the timed bodies costs 60-80 ms)
torch.cuda.synchronize()
import time
t0 = time.time()
host = []
for _ in range(n):
    t1 = time.perf_counter()
    eng.train_step(real, audio, slices, inputs_ready=ready)
    host.append((time.perf_counter() - t1) * 1e3)
t_host = time.time() - t0
torch.cuda.synchronize()
print("ms/body %.3f   host enqueue loop %.3f ms/body (per call: %s)" % (
    (time.time() - t0) * 1e3 / n, t_host * 1e3 / n, " ".join("%.1f" % h for h in host)))
if os.environ.get("HOST_ONLY"):
    # how long does the host need when it never has to wait for the device? (queue drained before every call)
    hs = []
    for _ in range(8):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.train_step(real, audio, slices, inputs_ready=ready)
        hs.append((time.perf_counter() - t1) * 1e3)
    print("host time per call on an empty queue: %s" % " ".join("%.1f" % h for h in hs))
