"""Dev tool: wall time per loop body of the REAL train scripts on a dataset folder (synthetic takes in the dataset's
on-disk format): batches gathered from the HBM-resident dataset (default) vs --host-loader (torch DataLoader + collate
on the host, the reference's way).   python tools/train_data_path.py [takes] [seconds per take] [bodies]"""
import os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from music2dance_amd.data import write_synthetic_dataset
from music2dance_amd.phase1 import train_wgan_gp as T1
from music2dance_amd.phase2 import train as T2
from music2dance_amd.phase3 import train as T3

takes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bodies = int(sys.argv[3]) if len(sys.argv) > 3 else 160
tmp = tempfile.mkdtemp(dir="/tmp")
folder = write_synthetic_dataset(os.path.join(tmp, "ds"), n_takes=takes, seconds=seconds, seed=4)
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "music2dance_amd")


def cfg(src, **over):
    c = yaml.safe_load(open(os.path.join(pkg, src)))
    c.update(over)
    path = os.path.join(tmp, "cfg.yaml")
    open(path, "w").write(yaml.safe_dump(c))
    return path


def timed(main, args, n):
    # the script is timed from its first loop body: wrap train_step
    import music2dance_amd.engine as E
    marks = {}
    for cls in (E.Phase1Engine, E.Phase2Engine, E.Phase3Engine):
        if "_orig_ts" not in cls.__dict__:
            cls._orig_ts = cls.train_step

            def ts(self, *a, _o=cls._orig_ts, **k):
                if self.total_iterations == n // 4 and "t0" not in marks:
                    torch.cuda.synchronize(); marks["t0"] = time.time(); marks["i0"] = self.total_iterations
                return _o(self, *a, **k)
            cls.train_step = ts
            cls._marks = marks
        else:
            cls._marks.clear(); marks = cls._marks
    eng = main(args + ["--iterations", str(n)])
    torch.cuda.synchronize()
    return (time.time() - marks["t0"]) / (eng.total_iterations - marks["i0"]) * 1e3


for name, main, src, over, extra in (
        ("phase 3, batch 64", T3.main, "phase3/configs/default.yaml", dict(batch_size=64, num_epochs=100000, folder=folder), []),
        ("phase 2, batch 32", T2.main, "phase2/configs/default.yaml", dict(batch_size=32, num_epochs=100000), ["--folder", folder]),
        ("phase 1, batch 64", T1.main, "phase1/configs/b2l50s32.yaml", dict(batch_size=64, num_epochs=100000), ["--folder", folder])):
    for mode in ([], ["--host-loader"]):
        c = cfg(src, **over)
        n = bodies if "phase 3" in name else bodies * 4
        ms = timed(main, ["-c", c, "-d", "0", "-n", "x", "--no-run-dir", "--log-every", "1000000"] + extra + mode, n)
        print("%-18s %-14s %8.2f ms per loop body" % (name, "host loader" if mode else "resident", ms), flush=True)
shutil.rmtree(tmp)
