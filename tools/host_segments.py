"""Dev tool: host-side (enqueue) time of the segments of a critic iteration, no device syncs."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels, losses, engine as E
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=1)
acc = collections.OrderedDict()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(eng.optim_critic, "zero_grad", "zero_grad")
wrap(eng.optim_critic, "step", "adam critic")
wrap(eng.optim_gen, "step", "adam gen")
wrap(E, "gradient_penalty", "gradient_penalty (fwd + 1st bwd)")
wrap(critic, "score_pair", "score_pair fwd")
orig_call = gen.forward
def gen_fwd(*a, **k):
    t = time.perf_counter(); r = orig_call(*a, **k); acc["gen forward"] = acc.get("gen forward", 0.0) + time.perf_counter() - t; return r
gen.forward = gen_fwd
orig_bw = torch.Tensor.backward
def bw(self, *a, **k):
    t = time.perf_counter(); r = orig_bw(self, *a, **k); acc["backward()"] = acc.get("backward()", 0.0) + time.perf_counter() - t; return r
torch.Tensor.backward = bw
for _ in range(8): eng.train_step(real, audio, slices)
torch.cuda.synchronize(); acc.clear()
N = 16
t0 = time.perf_counter()
for _ in range(N): eng.train_step(real, audio, slices)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host total per step %.2f ms, until device idle %.2f ms" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
for k, v in acc.items(): print("%-34s %.2f ms/step" % (k, 1e3 * v / N))
