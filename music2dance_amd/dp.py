"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The WGAN-GP step shards by sample: the critic has no cross-sample
coupling and its loss is a mean of per-sample terms, so summing per-rank gradients and
dividing by the world size reproduces the global-batch gradient; the generator does the
same with per-rank BatchNorm statistics. The only exchange step is this all-reduce.

Gradients are packed into a few persistent flat fp32 buckets (xGMI is point-to-point: few
large messages beat many small ones) with one fused multi-tensor copy per bucket and reduced
on a side stream, so that the next critic iteration's generator forward, which does not
depend on the critic's weights, overlaps the exchange; afterwards the parameters' .grad are
views of the buckets (no unpack). With `overlap_backward()` armed (the critic: 41.7 MB every
iteration) a bucket's all-reduce is launched from a post-accumulate-grad hook as soon as the
backward pass has produced its last gradient, i.e. underneath the rest of that backward; what
is still unlaunched when start() is called goes then. The generator's exchange (18.9 MB every
8th iteration, 4 MB buckets) is armed the same way: its one backward pass per generator iteration
launches every bucket but the last underneath itself, and start() / finish() right after the pass
wait for what is left (a ring over one 153 GB/s xGMI link moves 4 MB in ~50 us).
"""
import contextlib

import torch
import torch.distributed as dist


class GradExchange:
    """All-reduce(mean) of a parameter list's gradients through a few persistent flat fp32 buckets.

    start():  per bucket ONE fused multi-tensor copy (`torch._foreach_copy_`) packs the gradients
              that exist into the bucket (parameters without a gradient - the dead fc1 / bn1 branch of
              LinearBlock - keep a zero slot and stay without one), then the bucket's all-reduce is
              launched asynchronously on a communication stream;
    finish(): waits, scales by 1 / world (RCCL: inside the collective, ReduceOp.AVG) and REBINDS each
              p.grad to its slice of the bucket - no unpack copies. The next backward replaces p.grad
              (the engines zero gradients with set_to_none=True), so the views never alias new data.
    Buckets follow reverse parameter order (~ the order backward produces gradients) and are sized
    for xGMI's point-to-point links: few large messages (default 16 MB) instead of many small ones.
    `force=True` runs the exchange even at world size 1 (the RCCL smoke test on a single GPU)."""

    def __init__(self, params, bucket_mb=16.0, group=None, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.force = bool(force) and dist.is_available() and dist.is_initialized()
        self.buckets = []  # lists of params
        cap = int(bucket_mb * (1 << 20) / 4)
        cur, n = [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)    # persistent bucket storage
        self._views = [None] * len(self.buckets)   # per bucket: the parameters' slices, shaped like them
        self._stream = None
        self._pending = None
        self._avg_in_collective = None
        # backward-overlapped launch (overlap_backward): per bucket, how many gradients the backward pass
        # delivers (learnt from the first exchange: dead branches never deliver) and how many have arrived
        self._bucket_of = {}
        self._expect = None
        self._arrived = None
        self._launched = None
        self._next = 0
        self._stream_marks = {}  # raw stream handle -> event re-recorded by the hooks
        self._marked = set()
        self._hooks = []
        self._suspended = 0
        self.launched_in_backward = 0  # diagnostics: buckets whose all-reduce left before start()
        # diagnostics: how long the consuming stream (GPU) / the host (CPU backends) waited in finish() - the part of an
        # exchange that did NOT hide under the backward pass / the next generator forward. Event pairs, read by
        # wait_stats() once the device has caught up; bounded.
        self._wait_pairs = []
        # one more float at the end of the LAST bucket: "a recurrent launch of this rank gave up" (kernels.fault_fetch
        # fills it in before the bucket leaves). Summed / averaged with the gradients it is > 0 on EVERY rank as soon as
        # one rank raised it: the flag all ranks' optimizers take as their step's `skip` word (fault_flag), so that the
        # step one rank must void is voided everywhere and the replicas stay identical.
        self.fault_fetch = None

    @property
    def active(self):
        return self.world > 1 or self.force

    def _comm_stream(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _bucket(self, i, device):
        if self._flat[i] is None:
            total = sum(p.numel() for p in self.buckets[i])
            flat = torch.zeros(total + (1 if i == len(self.buckets) - 1 else 0), dtype=torch.float32, device=device)
            views, off = [], 0
            for p in self.buckets[i]:
                views.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._flat[i], self._views[i] = flat, views
        return self._flat[i], self._views[i]

    def _launch(self, i):
        """Pack bucket i on the current stream and start its all-reduce on the communication stream."""
        bucket = self.buckets[i]
        device = self.params[0].device
        stream = self._comm_stream(device)
        if self._avg_in_collective is None:
            # RCCL averages inside the collective; gloo (CPU tests, single-GPU dry runs) only sums
            self._avg_in_collective = device.type == "cuda" and dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        flat, views = self._bucket(i, device)
        src = [p.grad for p in bucket if p.grad is not None]
        dst = [v for p, v in zip(bucket, views) if p.grad is not None]
        aliased = bool(src) and all(s.data_ptr() == d.data_ptr() for s, d in zip(src, dst))
        if len(src) != len(bucket) and not aliased:
            flat.zero_()  # slots of parameters without a gradient contribute zeros
        if src and not aliased:
            torch._foreach_copy_(dst, src)
        if i == len(self.buckets) - 1:
            if self.fault_fetch is not None:
                self.fault_fetch(flat[-1:])
            else:
                flat[-1:].zero_()
        had = [p.grad is not None for p in bucket]
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(stream):
                work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        return work, had

    def fault_flag(self, fetch):
        """Arm the fault slot: `fetch(dst)` writes this rank's state (1.0 / 0.0) into the one-element device tensor dst
        on the current stream. -> that tensor: after finish() it holds the reduced value (> 0: some rank raised it)."""
        self.fault_fetch = fetch
        flat, _ = self._bucket(len(self.buckets) - 1, self.params[0].device)
        return flat[-1:]

    def overlap_backward(self):
        """Arm the backward-overlapped launch: from now on a bucket's all-reduce starts inside backward(),
        from the hook of the last gradient it waits for. The first exchange runs as before and records which
        parameters receive gradients at all; a bucket whose expected gradients do not all arrive is launched by
        start() like the rest. Gradients must be complete when their hook fires, i.e. ONE backward pass per
        exchange (the engines' critic iteration), and every rank must see the same set of live parameters."""
        if self._hooks:
            return self
        for i, bucket in enumerate(self.buckets):
            for p in bucket:
                self._bucket_of[id(p)] = i
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    @contextlib.contextmanager
    def suspended(self):
        """Inside: backward passes do not feed the exchange (no arrival counting, no hook-launched all-reduce).
        For a backward that is NOT followed by start() - a captured-graph warm-up pass, a diagnostic backward:
        its hooks would otherwise pack and reduce THOSE gradients, and the next start() would adopt the stale
        works instead of launching the real ones."""
        self._suspended += 1
        try:
            yield self
        finally:
            self._suspended -= 1

    def _on_grad(self, p):
        if self._expect is None or self._pending is not None or self._suspended:
            return
        if p.is_cuda and torch.cuda.is_current_stream_capturing():
            return  # captured backward passes: the exchange stays outside the graph
        i = self._bucket_of[id(p)]
        self._arrived[i] += 1
        if p.is_cuda:
            # a leaf's gradient is accumulated on the stream its accumulator node was created on (the critic's pose
            # branch lives on a side stream): remember where each stream stands now; the launch below, which
            # packs the whole bucket from ONE hook, first waits for all of them
            st = torch.cuda.current_stream(p.device)
            ev = self._stream_marks.get(st.cuda_stream)
            if ev is None:
                ev = self._stream_marks[st.cuda_stream] = torch.cuda.Event()
            ev.record(st)
            self._marked.add(st.cuda_stream)
        # collectives must be issued in the same order on every rank: strictly by bucket index
        # (a bucket that expects no gradient at all - every parameter dead - goes as soon as its turn comes)
        while self._next < len(self.buckets) and self._arrived[self._next] >= self._expect[self._next]:
            if p.is_cuda:
                cur = torch.cuda.current_stream(p.device)
                for key in self._marked:
                    if key != cur.cuda_stream:
                        cur.wait_event(self._stream_marks[key])
            self._launched[self._next] = self._launch(self._next)
            self.launched_in_backward += 1
            self._next += 1

    def poll(self):
        """For schedules that bind p.grad themselves (critic_step.CriticStep: no autograd hooks fire): launch, strictly
        in bucket order, every bucket whose expected gradients are all in place. Same contract as the hooks - armed
        by overlap_backward(), expectations learnt from the first exchange, what is left goes at start()."""
        if not self.active or not self._hooks or self._expect is None or self._pending is not None or self._suspended:
            return
        while self._next < len(self.buckets):
            i = self._next
            if sum(1 for p in self.buckets[i] if p.grad is not None) < self._expect[i]:
                break
            self._launched[i] = self._launch(i)
            self.launched_in_backward += 1
            self._next += 1

    def start(self):
        """Pack the gradients and launch the all-reduces asynchronously (those not already launched from
        the backward hooks)."""
        if not self.active:
            return
        assert self._pending is None, "previous exchange not finished"
        works, had = [], []
        for i in range(len(self.buckets)):
            done = self._launched[i] if self._launched is not None else None
            w, h = done if done is not None else self._launch(i)
            works.append(w)
            had.append(h)
        if self._hooks:
            # what the next backward is expected to deliver per bucket; arrival counters re-armed
            self._expect = [sum(h) for h in had]
            self._arrived = [0] * len(self.buckets)
            self._launched = [None] * len(self.buckets)
            self._next = 0
            self._marked = set()
        self._pending = (works, had)

    def finish(self):
        """Wait for the exchange; p.grad becomes the averaged gradient (a view of its bucket)."""
        if not self.active or self._pending is None:
            return
        works, had = self._pending
        device = self.params[0].device
        cur = torch.cuda.current_stream(device) if device.type == "cuda" else None
        capturing = cur is not None and torch.cuda.is_current_stream_capturing()
        if cur is not None and not capturing:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        else:
            import time
            t0 = time.perf_counter()
        for w in works:
            w.wait()
        if self._stream is not None:
            cur.wait_stream(self._stream)
        if cur is not None and not capturing:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(cur)   # e0 -> e1 on the consuming stream holds nothing but the waits
            self._wait_pairs.append((e0, e1))
        elif cur is None:
            self._wait_pairs.append(1e3 * (time.perf_counter() - t0))
        if len(self._wait_pairs) > 512:
            del self._wait_pairs[:256]
        inv = 1.0 / self.world
        for flat, views, bucket, has in zip(self._flat, self._views, self.buckets, had):
            if not self._avg_in_collective and self.world > 1:
                flat.mul_(inv)
            for p, v, h in zip(bucket, views, has):
                if h:
                    p.grad = v
        self._pending = None

    def exchange(self):
        self.start()
        self.finish()

    def wait_stats(self, reset=False):
        """-> {"exchanges", "wait_ms_mean", "wait_ms_max"}: time finish() blocked its stream (host, on CPU backends) per
        exchange since the last reset. Call with the device idle (event times are read); unfinished pairs are skipped."""
        vals = []
        for p in self._wait_pairs:
            if isinstance(p, float):
                vals.append(p)
            elif p[1].query():
                vals.append(p[0].elapsed_time(p[1]))
        if reset:
            self._wait_pairs = []
        if not vals:
            return {"exchanges": 0, "wait_ms_mean": None, "wait_ms_max": None}
        return {"exchanges": len(vals), "wait_ms_mean": round(sum(vals) / len(vals), 4), "wait_ms_max": round(max(vals), 4)}


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
