"""Dev tool: which lines of the package launch the non-m2d (ATen / runtime) kernels of a phase-3 loop body?
torch.profiler with Python stacks over 8 bodies (one generator iteration); per ATen op that reaches the device, the
innermost music2dance_amd frame and the count per body."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=1)
for _ in range(16): eng.train_step(real, audio, slices)
torch.cuda.synchronize()
BODIES = 8
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(BODIES): eng.train_step(real, audio, slices)
    torch.cuda.synchronize()
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::neg",
         "aten::mean", "aten::sub", "aten::div", "aten::_foreach", "aten::_fused", "aten::contiguous", "aten::clone")
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"): continue
    if not ev.kernels: continue
    site = "?"
    for fr in ev.stack:
        if "music2dance_amd" in fr or "bench.py" in fr:
            site = fr.split("music2dance_amd/")[-1]
            break
    cnt[(ev.name, site, ev.kernels[0].name[:60])] += 1
for (name, site, kern), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%6.2f/body  %-22s %-60s %s" % (n / BODIES, name, site, kern))
