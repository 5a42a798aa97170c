"""Dev tool: source lines of the package that launch ATen kernels in a steady-state phase-3 loop body (TorchDispatchMode:
every aten op that runs on the GPU outside the engine's own kernels, with the innermost music2dance_amd frame)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev)
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(64, 120, dev, seed=1)
for _ in range(16): eng.train_step(real, audio, slices)
torch.cuda.synchronize()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.alias", "aten.reshape", "aten.empty", "aten.unfold",
        "aten.is_", "aten.size", "aten.stride", "aten._local_scalar", "aten.lift_fresh", "aten.record_stream", "aten.set_", "aten.unbind",
        "aten.split", "aten.chunk", "aten.narrow", "aten.zeros", "aten.ones", "aten.resize_", "aten.new_empty", "aten.empty_like", "aten._to_copy")
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if not name.startswith(SKIP) or name.startswith("aten._to_copy"):
            ts = [a for a in list(args) + [out] if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts):
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "music2dance_amd/" in fr.filename and "tools/" not in fr.filename:
                        site = "%s:%d" % (fr.filename.split("music2dance_amd/")[-1], fr.lineno)
                        break
                cnt[(name, site)] += 1
        return out
BODIES = 8
with Mode():
    for _ in range(BODIES): eng.train_step(real, audio, slices)
torch.cuda.synchronize()
for (name, site), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print("%6.2f/body  %-34s %s" % (n / BODIES, name, site))
