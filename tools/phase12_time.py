"""Dev tool: wall time per loop body of the phase-2 and phase-1 train scripts (synthetic data, GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music2dance_amd.phase2 import train as t2
from music2dance_amd.phase1 import train_wgan_gp as t1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for mod, cfgdir in ((t2, "phase2"), (t1, "phase1")):
    d = os.path.join(os.path.dirname(mod.__file__), "configs")
    cfg = os.path.join(d, sorted(os.listdir(d))[0])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        mod.main(["-c", cfg, "-d", "0", "-n", "t", "--synthetic", "--iterations", str(n), "--no-run-dir", "--log-every", "100000"])
        torch.cuda.synchronize()
        print(cfgdir, os.path.basename(cfg), "rep", rep, "%.2f ms per loop body" % ((time.time() - t0) * 1e3 / n), flush=True)
