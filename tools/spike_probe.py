"""Dev tool: where do the occasional 30-50 ms loop bodies come from? Per body: GPU time between end-of-body
events, host time of the train_step call, garbage collections (generation, duration) and new device
allocations that happened inside the call.   python tools/spike_probe.py [steps] [gc: on|off|freeze]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "on"
dev = torch.device("cuda", 0)
gen, critic = bench.build_models(dev, 120)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=100)
gen.train(); critic.train()
torch.cuda.synchronize()
ready = torch.cuda.current_stream().record_event()
torch.manual_seed(1234)
for _ in range(16):
    eng.train_step(real, audio, slices, inputs_ready=ready)
eng.flush(); torch.cuda.synchronize()
if mode == "off":
    gc.disable()
elif mode == "freeze":
    gc.collect(); gc.freeze()
log = []
gcs = []
t_gc = [0.0]
def cb(phase, info):
    if phase == "start": t_gc[0] = time.perf_counter()
    else: gcs.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))
gc.callbacks.append(cb)
evs = []
for i in range(steps):
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    g0 = len(gcs)
    t0 = time.perf_counter()
    eng.train_step(real, audio, slices, inputs_ready=ready)
    host = (time.perf_counter() - t0) * 1e3
    ev = torch.cuda.Event(enable_timing=True); ev.record(); evs.append(ev)
    log.append((host, torch.cuda.memory_stats().get("num_device_alloc", 0) - a0, gcs[g0:]))
eng.flush(); torch.cuda.synchronize()
gpu = [0.0] + [a.elapsed_time(b) for a, b in zip(evs, evs[1:])]
print("mode", mode, "mean gpu %.2f ms" % (sum(gpu[1:]) / (len(gpu) - 1)), "mean host %.2f ms" % (sum(l[0] for l in log) / len(log)))
for i, (g, (h, na, gl)) in enumerate(zip(gpu, log)):
    if g > 20 and not (22 < g < 26) or h > 20 or na or any(d > 3 for _, d in gl):
        print("  step %3d gpu %6.1f ms host %6.1f ms new_allocs %d gcs %s" % (i, g, h, na, [(a, round(d, 1)) for a, d in gl]))
