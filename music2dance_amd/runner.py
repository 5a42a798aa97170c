"""Shared scaffolding of the three train scripts: YAML config, run folders, optional
TensorBoard, checkpoints with the reference's file names, synthetic batches.

The reference's dataset pipeline (JSON skeletons, wav loading, samplers: utils.py:15-194,
phase3/train.py:114-162) is outside this engine's scope: the scripts
run on `--synthetic` batches of the dataset's shapes, or on tensors a caller provides.
"""
import datetime
import os

import torch
import yaml


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def pick_device(idx):
    if not torch.cuda.is_available():
        raise RuntimeError("music2dance_amd trains on an MI355X: no HIP device is visible and there is no CPU path")
    return torch.device("cuda:" + str(0 if idx is None else idx))


def settle_garbage_collector():
    """Call once the models / engine are built: collect, then move everything alive into the permanent
    generation. A full (generation-2) collection over the module / parameter / closure graph of a training
    process takes 60-80 ms of host time (measured, tools/spike_probe.py) - five loop bodies during which the
    launch queue runs dry; afterwards a full collection only walks what the loop itself allocates."""
    import gc
    gc.collect()
    gc.freeze()


def make_run_dir(name, enabled=True):
    if not enabled:
        return None
    os.makedirs("./runs", exist_ok=True)
    logdir = "./runs/" + datetime.datetime.now().strftime("%Y%m%d-%H%M%S") + "_" + str(name)
    os.makedirs(logdir)
    os.makedirs(logdir + "/samples")
    os.makedirs(logdir + "/models")
    return logdir


class ScalarLog:
    """TensorBoard scalars under the reference's tags when tensorboard is importable;
    otherwise a no-op. Values stay on the device until `every` iterations have passed, so
    throughput runs do not pay one host sync per scalar per iteration."""

    def __init__(self, logdir, every=1):
        self.every = max(int(every), 1)
        self.writer = None
        self.last = {}  # most recent value per tag (device tensors; tests and prints read it)
        if logdir is not None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(logdir + "/logging")
            except Exception:
                self.writer = None

    def scalars(self, values, step, force=False):
        self.last.update(values)
        if self.writer is None or (step % self.every and not force):
            return
        for tag, v in values.items():
            self.writer.add_scalar(tag, float(v), step)


def dump_architectures(logdir, gen, critic):
    if logdir is None:
        return
    with open(logdir + "/model_gen.txt", "w+") as f:
        f.write(str(gen))
    with open(logdir + "/model_critic.txt", "w+") as f:
        f.write(str(critic))


def save_state(module, path):
    torch.save(module.state_dict(), path)
