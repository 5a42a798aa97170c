"""Dev tool: cProfile of the host side of critic / generator iterations."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=1)
for _ in range(8): eng.train_step(real, audio, slices)
torch.cuda.synchronize()
import gc
if os.environ.get('NOGC'):
    gc.disable()
pr = cProfile.Profile()
pr.enable()
for _ in range(16): eng.train_step(real, audio, slices)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(14); print(s.getvalue()[:3000])
