"""Dev tool: cProfile of the launch thread over phase-2 loop bodies in captured-graph mode (GRAPHS=0: eager)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd import runner
from music2dance_amd.engine import Phase2Engine
from music2dance_amd.phase2.archis import default as p2
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = p2.SequenceGenerator(50, 50, 256, 69, 2, 3, dev)
critic = p2.SequenceDiscriminator(69, 128, 120, 25, 3, dev)
eng = Phase2Engine(gen, critic, bench.P2_DEFAULT)
if os.environ.get("GRAPHS", "1") != "0":
    eng.enable_graphs()
real = torch.rand(32, 120, 69, generator=torch.Generator().manual_seed(4)).to(dev)
for _ in range(24): eng.train_step(real)
torch.cuda.synchronize()
runner.settle_garbage_collector()
pr = cProfile.Profile()
pr.enable()
for _ in range(64):
    eng.train_step(real)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
