import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1])
torch.set_num_threads(n)
import bench
from oracle import m2d_oracle as O
from music2dance_amd.engine import synthetic_phase3_batch
gen, critic = bench.build_models("cpu")
gsd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
dsd = {k: v.detach().clone() for k, v in critic.state_dict().items()}
real, audio, slices = synthetic_phase3_batch(8, 120, "cpu", seed=1)
cfg = O.P3Config(n_critic=3)
t = time.perf_counter(); O.p3_train_iterations(gsd, dsd, cfg, real, audio, slices, 1, 0); t1 = time.perf_counter()
O.p3_train_iterations(gsd, dsd, cfg, real, audio, slices, 1, 0); t2 = time.perf_counter()
print("threads", n, "critic iter B=8: first %.2f s second %.2f s" % (t1 - t, t2 - t1), flush=True)
