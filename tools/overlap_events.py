"""Dev tool: where does the generator forward of a critic iteration sit relative to the critic's own work, WITHOUT a
profiler attached (rocprofv3's API interception slows the launch thread enough to change the picture)? HIP events
around the generator forward (its stream) and around CriticStep.run (main stream), 16 steady bodies."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
import bench
dev = torch.device("cuda:0")
_dummies = []
for _i in range(int(os.environ.get("STREAM_SHIFT", "0"))):  # see tools/steady.py
    _s = torch.cuda.Stream(dev)
    with torch.cuda.stream(_s):
        torch.zeros(8, device=dev)
    _dummies.append(_s)
torch.cuda.synchronize()
cfg = bench.PRESETS[os.environ.get("CONFIG", "c3")]
gen, critic = bench.build_models(dev, cfg["frames"], cfg["enc_type"], cfg["ablated"])
eng = Phase3Engine(gen, critic, bench.P3_DEFAULT, ablated=cfg["ablated"])
real, audio, slices, ready = synthetic_phase3_batch(cfg["batch"], cfg["frames"], dev, seed=1, with_event=True)
for _ in range(16): eng.train_step(real, audio, slices, inputs_ready=ready)
from music2dance_amd import runner
runner.settle_garbage_collector()  # (a generation-2 collection inside the timed bodies costs 60-80 ms)
torch.cuda.synchronize()
marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
gfwd = gen.forward
def gen_forward(*a, **k):
    s = ev(); out = gfwd(*a, **k); marks.append(("gen", s, ev(), torch.is_grad_enabled())); return out
gen.forward = gen_forward
crun = eng.manual_critic.run
def critic_run(*a, **k):
    s = ev(); out = crun(*a, **k); marks.append(("critic", s, ev(), False)); return out
eng.manual_critic.run = critic_run
# at every join of the critic's two streams: which one arrives last, and by how much?
joins = []
from music2dance_amd import critic_step as _cs
_join0 = _cs.CriticStep._join
def _join(side, cur, *tensors):
    if side is not None:
        es = torch.cuda.Event(enable_timing=True); ec = torch.cuda.Event(enable_timing=True)
        es.record(side); ec.record(cur)
        joins.append((es, ec))
    return _join0(side, cur, *tensors)
_cs.CriticStep._join = staticmethod(_join)
giter = eng.generator_iteration
def gen_iter(*a, **k):
    s = ev(); out = giter(*a, **k); marks.append(("gen-iteration", s, ev(), True)); return out
eng.generator_iteration = gen_iter
base = ev()
N = 16
for _ in range(N): eng.train_step(real, audio, slices, inputs_ready=ready)
end = ev()
torch.cuda.synchronize()
print("ms/body %.3f" % (base.elapsed_time(end) / N))
if os.environ.get("BRIEF"):
    marks = marks[8:28]
for name, s, e, grad in marks:
    print("%-14s %s %9.3f -> %9.3f  (%6.3f ms)" % (name, "grad" if grad else "    ", base.elapsed_time(s), base.elapsed_time(e), s.elapsed_time(e)))
if joins:
    import collections
    per = collections.defaultdict(list)
    nj = len(joins) // N
    for i, (es, ec) in enumerate(joins):
        per[i % nj].append(ec.elapsed_time(es))  # > 0: the side (pose) stream arrives after the main (audio) stream
    print("joins per critic pass: %d; side-stream arrival minus main-stream arrival (ms, mean over %d bodies):" % (nj, N))
    for j in range(nj):
        v = per[j]
        print("  join %d: %+.3f" % (j, sum(v) / len(v)))
