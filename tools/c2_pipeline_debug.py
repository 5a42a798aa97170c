"""Dev tool: phase-2 loop bodies with GPU-side events - when does the generator forward of iteration k start relative
to the critic pass of iteration k - 1? GRAPHS=0/1. Prints per body: gen-forward start / end and critic start / end
(us, relative to the critic start of that body), and how often the engine re-armed its parameters-ready event."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd.engine import Phase2Engine, WganGpEngine
from music2dance_amd.phase2.archis.default import SequenceDiscriminator as D2, SequenceGenerator as G2
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = G2(50, 50, 256, 69, 2, 3, dev); critic = D2(69, 128, 120, 25, 3, dev)
eng = Phase2Engine(gen, critic, bench.P2_DEFAULT); eng.host_noise = False
if os.environ.get("GRAPHS", "1") == "1": eng.enable_graphs()
real = torch.rand(32, 120, 69, generator=torch.Generator().manual_seed(100)).to(dev)
for _ in range(24): eng.train_step(real)
torch.cuda.synchronize()
rearm = [0]
orig = WganGpEngine._generator_forward_nograd
marks = []
def wrapped(self, fn, inputs, device=None):
    v0 = self._gen_params_ready
    def timed():
        gs = torch.cuda.current_stream(dev)
        e0 = torch.cuda.Event(enable_timing=True); e0.record(gs)
        out = fn()
        e1 = torch.cuda.Event(enable_timing=True); e1.record(gs)
        marks.append(("gen", e0, e1))
        return out
    out = orig(self, timed, inputs, device)
    if self._gen_params_ready is not v0: rearm[0] += 1
    return out
WganGpEngine._generator_forward_nograd = wrapped
ref = torch.cuda.Event(enable_timing=True)
N = 17
steps = []
for i in range(N):
    main = torch.cuda.current_stream(dev)
    if i == 0: ref.record(main)
    eng.train_step(real)
    e = torch.cuda.Event(enable_timing=True); e.record(main)
    steps.append(e)
torch.cuda.synchronize()
print("parameters-ready event re-armed in %d of %d bodies" % (rearm[0], N))
prev = 0.0
for i in range(N):
    t_end = ref.elapsed_time(steps[i]) * 1e3
    g0 = ref.elapsed_time(marks[i][1]) * 1e3; g1 = ref.elapsed_time(marks[i][2]) * 1e3
    print("body %2d: main-stream end %8.1f us (+%7.1f) | its generator forward ran %8.1f .. %8.1f" % (i, t_end, t_end - prev, g0, g1))
    prev = t_end
