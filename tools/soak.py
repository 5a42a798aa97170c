"""Dev tool: a few hundred loop bodies; checks finiteness and that device memory stays flat."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
batches = [synthetic_phase3_batch(64, 120, dev, seed=i) for i in range(4)]
torch.manual_seed(0)
mem = []
t0 = time.perf_counter()
for i in range(int(os.environ.get("STEPS", 400))):
    out = eng.train_step(*batches[i % 4])
    if i % 50 == 49:
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items()}
        assert all(math.isfinite(v) for v in vals.values()), vals
        mem.append(torch.cuda.memory_reserved() / 2**30)
        print(i + 1, "steps %.1f s" % (time.perf_counter() - t0), {k: round(v, 3) for k, v in vals.items()}, "reserved %.2f GiB" % mem[-1], flush=True)
eng.flush()
assert mem[-1] <= mem[1] * 1.05 + 0.1, mem
print("OK", mem)
