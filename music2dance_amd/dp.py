"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The WGAN-GP step shards by sample: the critic has no cross-sample
coupling and its loss is a mean of per-sample terms, so summing per-rank gradients and
dividing by the world size reproduces the global-batch gradient; the generator does the
same with per-rank BatchNorm statistics. The only exchange step is this all-reduce.

Gradients are packed into a few flat fp32 buckets (xGMI is point-to-point: few large
messages beat many small ones) and reduced on a side stream so that the next critic
iteration's generator forward, which does not depend on the critic's weights, overlaps
the exchange.
"""
import torch
import torch.distributed as dist


class GradExchange:
    def __init__(self, params, bucket_mb=16.0, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = []  # lists of params
        cap = int(bucket_mb * (1 << 20) / 4)
        cur, n = [], 0
        for p in reversed(self.params):  # reverse order ~ the order backward produces gradients
            if cur and n + p.numel() > cap:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._stream = None
        self._pending = None

    @property
    def active(self):
        return self.world > 1

    def _comm_stream(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def start(self):
        """Pack gradients and launch the all-reduces asynchronously. Parameters without a
        gradient (the dead fc1 / bn1 branch) contribute zeros and are left without one."""
        if not self.active:
            return
        assert self._pending is None, "previous exchange not finished"
        device = self.params[0].device
        stream = self._comm_stream(device)
        works = []
        for i, bucket in enumerate(self.buckets):
            total = sum(p.numel() for p in bucket)
            flat = self._flat[i]
            if flat is None or flat.numel() != total:
                flat = torch.empty(total, dtype=torch.float32, device=device)
                self._flat[i] = flat
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            if stream is not None:
                stream.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(stream):
                    works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._pending = works

    def finish(self):
        """Wait for the exchange and write the averaged gradients back."""
        if not self.active or self._pending is None:
            return
        for w in self._pending:
            w.wait()
        device = self.params[0].device
        if self._stream is not None:
            torch.cuda.current_stream(device).wait_stream(self._stream)
        inv = 1.0 / self.world
        for flat, bucket in zip(self._flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad)).mul_(inv)
                off += n
        self._pending = None

    def exchange(self):
        self.start()
        self.finish()


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* when
    launched by torch.distributed.run; returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
