"""Dev tool: does it pay to run the loop body's MAIN stream at high priority (side streams - generator forward, pose branch -
at the default, lower one), so that the side streams' workgroups only take what the main chain's launches leave free?
Prints ms per loop body for both arrangements, alternating (one box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd import runner
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch

dev = torch.device("cuda:0")
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
gen, critic = bench.build_models(dev, 120)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=100)
gen.train(), critic.train()
torch.cuda.synchronize()
hi = torch.cuda.Stream(device=dev, priority=-1)


def run(stream, steps=24, warm=8):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
    with ctx:
        ready = torch.cuda.current_stream(dev).record_event()
        for _ in range(warm):
            eng.train_step(real, audio, slices, inputs_ready=ready)
        eng.flush()
        runner.settle_garbage_collector()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(real, audio, slices, inputs_ready=ready)
        eng.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


for r in range(3):
    print("round %d: main on the default stream %.3f ms | main on a high-priority stream %.3f ms" % (r, run(None), run(hi)), flush=True)
