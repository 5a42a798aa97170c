"""Dev tool: what does the data-parallel path cost per loop body on ONE GPU? RCCL at world size 1 with the gradient
exchange forced on (bucket packing, the communication stream, the all-reduce launches, the deferred critic step) next to
the same engine without it - the part of N-GPU scaling loss that does not come from the wire."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
from music2dance_amd import runner
from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
dev = torch.device("cuda:0")
real, audio, slices, ready = synthetic_phase3_batch(64, 120, dev, seed=1, with_event=True)
for forced in (False, True, False, True):
    gen, critic = bench.build_models(dev)
    eng = Phase3Engine(gen, critic, bench.P3_DEFAULT)
    if forced:
        eng.x_critic.force = eng.x_gen.force = True
    for _ in range(16): eng.train_step(real, audio, slices, inputs_ready=ready)
    eng.flush(); torch.cuda.synchronize()
    runner.settle_garbage_collector()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 24
    for _ in range(N): eng.train_step(real, audio, slices, inputs_ready=ready)
    eng.flush()
    e1.record(); torch.cuda.synchronize()
    print("exchange %s: %.3f ms per body" % ("FORCED (RCCL, world 1)" if forced else "off", e0.elapsed_time(e1) / N), flush=True)
    del eng, gen, critic
dist.destroy_process_group()
