"""Dev tool: do repeated runs of the train scripts on a dataset folder end in identical parameters? Pairs of
(resident, resident), (host, host), (resident, host) runs of each phase; prints the worst parameter difference."""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from music2dance_amd.data import write_synthetic_dataset
from music2dance_amd.phase1 import train_wgan_gp as T1
from music2dance_amd.phase2 import train as T2
from music2dance_amd.phase3 import train as T3

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tmp = tempfile.mkdtemp(dir="/tmp")
folder = write_synthetic_dataset(os.path.join(tmp, "ds"), n_takes=12, seconds=6, seed=2)
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "music2dance_amd")
os.chdir(tmp)


def cfg(src, **over):
    c = yaml.safe_load(open(os.path.join(pkg, src)))
    c.update(over)
    path = os.path.join(tmp, "cfg.yaml")
    open(path, "w").write(yaml.safe_dump(c))
    return path


def run(phase, host):
    extra = ["--host-loader"] if host else []
    if phase == 3:
        e = T3.main(["-c", cfg("phase3/configs/default.yaml", batch_size=4, num_epochs=2, n_critic_steps=2, folder=folder),
                     "-d", "0", "-n", "r3", "--no-run-dir"] + extra)
    elif phase == 2:
        e = T2.main(["-c", cfg("phase2/configs/default.yaml", batch_size=4, num_train=10, num_epochs=2, n_critic_steps=2),
                     "-d", "0", "-n", "r2", "--no-run-dir", "--folder", folder] + extra)
    else:
        e = T1.main(["-c", cfg("phase1/configs/b2l50s32.yaml", batch_size=8, num_train=40, num_epochs=1),
                     "-d", "0", "-n", "r1", "--no-run-dir", "--folder", folder] + extra)
    torch.cuda.synchronize()
    return [p.detach().clone() for m in (e.gen, e.critic) for p in m.parameters()]


import io, contextlib
import numpy as np
for phase in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '3,2,1').split(',')]:
    ref = None
    for r in range(reps):
        for host in (False, True):
            torch.manual_seed(0); np.random.seed(3)  # (phases 1 / 2 seed nothing themselves, like the reference)
            with contextlib.redirect_stdout(io.StringIO()):
                p = run(phase, host)
            if ref is None:
                ref = p
                continue
            worst = max(float((a - b).abs().max()) for a, b in zip(p, ref))
            nbad = sum(1 for a, b in zip(p, ref) if not torch.equal(a, b))
            if worst > 0:
                print("phase %d rep %d %s: %d tensors differ from the first run, worst %.3e" % (phase, r, "host" if host else "resident", nbad, worst), flush=True)
    print("phase %d: %d runs compared" % (phase, 2 * reps - 1), flush=True)
shutil.rmtree(tmp)
