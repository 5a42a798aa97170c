"""Dev tool: the hand-scheduled critic iteration on the HIP kernels against THE SAME schedule evaluated in fp64 on the
host (tests/fake_backend.py on double tensors), intermediate by intermediate (BASELINE configs[2] critic)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from music2dance_amd import kernels
from music2dance_amd.critic_step import CriticStep
from tests.fake_backend import FakeKernels
from tests.test_critic_step import _inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with_audio = len(sys.argv) <= 2
dev = torch.device("cuda:0")
_, critic = bench.build_models(dev, 120, ablated=not with_audio)
real, fake_rows, alpha, audio = _inputs(B, 120, dev, seed=9, audio=with_audio)
torch.set_num_threads(min(64, os.cpu_count() or 8))
c64 = copy.deepcopy(critic).cpu().double()
step = CriticStep(critic, 10.0)
step.debug = {}
with kernels.impl().weight_cache():
    out = step.run(real, fake_rows, None if audio is None else audio.clone(), alpha)
torch.cuda.synchronize()
prev = kernels.set_impl(FakeKernels())
s64 = CriticStep(c64, 10.0)
s64.debug = {}
out64 = s64.run(real.cpu().double(), fake_rows.cpu().double(), None if audio is None else audio.cpu().double(), alpha.cpu().double())
kernels.set_impl(prev)
for k in step.debug:
    a, b = step.debug[k].cpu().double(), s64.debug[k]
    sc = b.abs().max().item()
    rows = a.shape[0] // 3 if a.shape[0] % 3 == 0 and a.shape[0] >= 3 else None
    extra = ""
    if rows and a.dim() >= 2:
        extra = "  interp %.1e pair %.1e" % ((a[:rows] - b[:rows]).abs().max().item() / sc, (a[rows:] - b[rows:]).abs().max().item() / sc)
    print("%-8s max %.3e  err %.2e%s  n(|d| > 1e-4 max) %d" % (k, sc, (a - b).abs().max().item() / sc, extra,
                                                           int(((a - b).abs() > 1e-4 * sc).sum())))
for (n, p), (_, p64) in zip(critic.named_parameters(), c64.named_parameters()):
    sc = p64.grad.abs().max().item()
    print("%-34s %10.3e %10.2e" % (n, sc, (p.grad.cpu().double() - p64.grad).abs().max().item() / max(sc, 1e-30)))
print({k: (float(out[k]), float(out64[k])) for k in out})
