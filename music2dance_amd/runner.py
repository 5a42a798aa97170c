"""Shared scaffolding of the three train scripts: YAML config, run folders, optional
TensorBoard, checkpoints with the reference's file names, synthetic batches.

The scripts run on `--synthetic` batches of the dataset's shapes, or - through music2dance_amd.data, the
reference's dataset pipeline (utils.py:15-194, phase3/train.py:114-162) - on a Music-to-Dance-Motion-Synthesis
folder (`folder:` of the YAML, or `--folder`).
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
    """TensorBoard scalars under the reference's tags when tensorboard is importable; otherwise a no-op.
    The reference writes `loss.item()` every generator iteration (phase3/train.py:228-236): a host sync per scalar that
    drains the launch queue. Here a logged value is copied to pinned host memory on the stream that made it
    (non-blocking) and WRITTEN when that copy is known to have finished - on a later `scalars()` call, or in
    `flush()` at the end of the run: same tags, steps and values, no stall in the loop."""

    def __init__(self, logdir, every=1, writer=None):
        self.every = max(int(every), 1)
        self.writer = writer
        self.last = {}  # most recent value per tag (device tensors; tests and prints read it)
        self._queue = []  # (tag, step, host tensor or float, event or None)
        if logdir is not None and writer is None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(logdir + "/logging")
            except Exception:
                self.writer = None

    def _drain(self, block=False):
        while self._queue:
            tag, step, host, ev = self._queue[0]
            if ev is not None and not block and not ev.query():
                return
            if ev is not None and block:
                ev.synchronize()
            self._queue.pop(0)
            self.writer.add_scalar(tag, float(host), step)

    def scalars(self, values, step, force=False):
        self.last.update(values)
        if self.writer is None or (step % self.every and not force):
            return
        for tag, v in values.items():
            if torch.is_tensor(v) and v.is_cuda:
                host = torch.empty((), dtype=v.dtype, pin_memory=True)
                host.copy_(v.detach().reshape(()), non_blocking=True)
                self._queue.append((tag, step, host, torch.cuda.current_stream(v.device).record_event()))
            else:
                self._queue.append((tag, step, v, None))
        self._drain()

    def flush(self):
        if self.writer is not None:
            self._drain(block=True)
            if hasattr(self.writer, "flush"):
                self.writer.flush()


def dump_architectures(logdir, gen, critic):
    if logdir is None:
        return
    with open(logdir + "/model_gen.txt", "w+") as f:
        f.write(str(gen))
    with open(logdir + "/model_critic.txt", "w+") as f:
        f.write(str(critic))


def save_state(module, path):
    torch.save(module.state_dict(), path)


def dataset_folder(cfg, override=None):
    """The dataset folder of a non-synthetic run (`--folder` or the YAML's `folder:` key); exits with the
    reference's own situation spelled out when it is absent (the data is not distributed with either repo)."""
    folder = override or cfg.get("folder")
    if not folder or not os.path.isdir(folder):
        raise SystemExit("dataset folder %r not found: pass --folder <Music-to-Dance-Motion-Synthesis-master> "
                         "(or --synthetic for random batches of the dataset's shapes)" % (folder,))
    return folder


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _tensors(y)


def resident_batches(loader, device, derive=None):
    """Batches of a data.ResidentLoader, each gathered on the copy stream: yields (batch, event after which its tensors
    are complete) like `staged` does for host batches - the gather of batch k + 1 runs beside the kernels of loop body k,
    and `train_step(..., inputs_ready=event)` may start its generator forward as soon as the gather is done.
    `derive(batch)` (optional) runs on the copy stream too, BEFORE the event: tensors computed from the batch that the
    step's other streams read (phase 3's padded audio track) - it replaces the batch in what is yielded."""
    from .layers import copy_stream
    it = iter(loader)
    cuda = torch.device(device).type == "cuda"
    while True:
        if not cuda:
            try:
                batch = next(it)
            except StopIteration:
                return
            yield (derive(batch) if derive else batch), None
            continue
        cur = torch.cuda.current_stream(device)
        cs = copy_stream(device)
        # (no wait for `cur`: the dataset is uploaded by the first gather on this same stream and never written again,
        # and the batch lands in fresh blocks of the copy stream's own pool)
        with torch.cuda.stream(cs):
            try:
                batch = next(it)
            except StopIteration:
                return
            if derive is not None:
                batch = derive(batch)
            ready = cs.record_event()
        cur.wait_event(ready)
        for t in _tensors(batch):
            t.record_stream(cur)
        yield batch, ready


def staged(tensors, device, derive=None):
    """Host tensors of one loader batch -> device, on the copy stream; (device tensors, event after which they
    are complete): what `train_step(..., inputs_ready=)` takes. `derive(device tensors)` (optional) runs on the copy
    stream before the event and replaces them: whatever else the step's other streams read must be complete at the
    event too (phase 3's padded audio track - made on the main stream it would sit behind the previous loop body
    while the generator stream, which only waits for the event, already reads it)."""
    from .layers import copy_stream
    if torch.device(device).type != "cuda":
        out = [t.to(device) for t in tensors]
        return (derive(out) if derive else out), None
    cur = torch.cuda.current_stream(device)
    cs = copy_stream(device)
    with torch.cuda.stream(cs):
        out = [t.to(device, non_blocking=True) for t in tensors]
        if derive is not None:
            out = derive(out)
        ready = cs.record_event()
    cur.wait_event(ready)
    for t in _tensors(out):
        t.record_stream(cur)
    return out, ready
