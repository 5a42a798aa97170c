"""Dev tool: soak run of the phase-3 trainer on the GPU box - N loop bodies of the real train script
(staged batches, pipelined generator forward, log every body), then: finite losses, no asynchronous
kernel error, allocator high-water mark flat over the second half of the run.
    python tools/soak.py [iterations] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd import kernels
from music2dance_amd.phase3 import train

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = os.path.join(os.path.dirname(train.__file__), "configs", "default.yaml")
marks = []
orig = train.Phase3Engine.train_step


def step(self, *a, **k):
    out = orig(self, *a, **k)
    if self.total_iterations % 50 == 0:
        torch.cuda.synchronize()
        marks.append((self.total_iterations, torch.cuda.max_memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20,
                      {kk: round(float(v), 4) for kk, v in out.items()}))
        print(marks[-1], flush=True)
    return out


train.Phase3Engine.train_step = step
t0 = time.time()
eng = train.main(["-c", cfg, "-d", "0", "-n", "soak", "--synthetic", "--iterations", str(n), "--batch-size", str(b),
                  "--no-run-dir", "--log-every", "50"])
torch.cuda.synchronize()
print("wall %.1f s for %d bodies" % (time.time() - t0, n))
kernels.impl().check_async_errors()
assert all(all(v == v and abs(v) < 1e30 for v in m[3].values()) for m in marks), "non-finite loss"
half = [m for m in marks if m[0] > n // 2]
assert half and half[-1][1] - half[0][1] <= 16, "allocator high-water mark still growing: %s" % ([m[:3] for m in half],)  # (MB)
print("soak ok")
