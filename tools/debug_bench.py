import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
from music2dance_amd import kernels
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
B = int(os.environ.get("B", 64))
dev = torch.device("cuda:0")
gen, critic = bench.build_models(dev, 120, os.environ.get("ENC", "default"))
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
real, audio, slices = synthetic_phase3_batch(B, 120, dev, seed=1)
print("built", flush=True)
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-28s %8.1f ms" % (name, 1e3 * (time.perf_counter() - t)), flush=True); return r
with torch.no_grad():
    T("gen fwd nograd", lambda: gen(slices, [120] * B))
    T("gen fwd nograd 2", lambda: gen(slices, [120] * B))
T("critic iter 1", lambda: eng.critic_iteration(real, audio, slices))
T("critic iter 2", lambda: eng.critic_iteration(real, audio, slices))
T("critic iter 3", lambda: eng.critic_iteration(real, audio, slices))
T("gen iter 1", lambda: eng.generator_iteration(real, audio, slices))
T("gen iter 2", lambda: eng.generator_iteration(real, audio, slices))
K = kernels.impl()
K.prof_begin()
T("critic iter prof", lambda: eng.critic_iteration(real, audio, slices))
eng.flush()
p = K.prof_end()
print({k: (round(v["ms"], 2), v["launches"], round(v["flops"] / 1e9, 1)) for k, v in p.items()}, flush=True)
K.prof_begin()
T("gen iter prof", lambda: eng.generator_iteration(real, audio, slices))
p = K.prof_end()
print({k: (round(v["ms"], 2), v["launches"], round(v["flops"] / 1e9, 1)) for k, v in p.items()}, flush=True)
import collections
for name, fn in (("critic", lambda: eng.critic_iteration(real, audio, slices)), ("gen", lambda: eng.generator_iteration(real, audio, slices))):
    K.prof_begin(); fn(); eng.flush(); torch.cuda.synchronize()
    rows = K.prof_dump(); K.prof_end()
    agg = collections.OrderedDict()
    for fam, tag, d0, d1, d2, ms, fl, _by in rows:
        if fam != 0 and not tag.startswith('thin'): continue
        k = (tag, d0, d1, d2)
        a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
    print("==== %s iteration: engine launches by shape (tag M N K: count, ms, TF/s)" % name)
    for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-24s %6d %8d %7d : %3d %8.3f ms %6.1f TF/s" % (k[0], k[1], k[2], k[3], c, ms, fl / ms / 1e9 if ms else 0))
# host enqueue time vs device time
import time
torch.cuda.synchronize()
for name, fn in (("critic", lambda: eng.critic_iteration(real, audio, slices)), ("gen", lambda: eng.generator_iteration(real, audio, slices))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: host enqueue %.1f ms, until device idle %.1f ms" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
