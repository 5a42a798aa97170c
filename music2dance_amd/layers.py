"""Parameter holders with the reference's names and initialisation, HIP forward.

Each class subclasses the torch.nn module the reference instantiates, so construction
consumes the RNG identically (seeded-constructor parity) and
state_dict keys / shapes are the reference's; only `forward` differs: it
dispatches to the gfx950 kernels through ops.py and can fuse the following activation.
"""
import contextlib

import torch
import torch.nn as nn

from . import ops



_COPY_STREAMS = {}


def copy_stream(device):
    """The per-device stream host -> device staging runs on: a HIP stream of the library's own (m2d_stream_create),
    wrapped as a torch ExternalStream - NOT one of torch's pooled streams. torch hands out its 32 pooled streams per
    priority round-robin, so the 33rd `torch.cuda.Stream()` of a process IS the first one again: same queue, same
    caching-allocator pool. A copy from pageable memory is written by the HOST once ITS stream has drained, i.e. it is not
    ordered against readers on other streams; with the copy stream aliasing a module's side stream, a gradient allocated on
    that side stream, read on the main stream (optimizer / clone) and freed could be handed to the next staging copy and
    overwritten while the main stream's read was still queued (seen in the test suite, which builds dozens of engines per
    process: the first 32 entries of a pose-branch bias gradient held the next batch's interpolation weights).
    (A pooled HIGH-priority stream avoids the aliasing too, but costs the phase-3 step 2 ms: 11.35 -> 13.49 per body.)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cs = _COPY_STREAMS.get(idx)
    if cs is None:
        import ctypes

        from . import _lib
        raw = ctypes.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib().m2d_stream_create(ctypes.byref(raw)), "m2d_stream_create")
        cs = _COPY_STREAMS[idx] = torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", idx))
    return cs


def to_device_async(t, device):
    """Move a HOST tensor (random draws from the host generator, as the reference makes them)
    to `device` without draining the compute stream. A copy from pageable memory blocks the
    host until everything queued before it on ITS stream has run; issued on the current stream
    that is a full host<->device sync per draw (the host then has to refill an empty queue).
    Here it runs on a dedicated copy stream - the host only waits for the copy itself - and the
    current stream picks the result up through an event."""
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    cs = copy_stream(device)
    cur = torch.cuda.current_stream(device)
    with torch.cuda.stream(cs):
        out = t.to(device)
    cur.wait_stream(cs)
    out.record_stream(cur)
    return out


class Conv1d(nn.Conv1d):
    """nn.Conv1d (dilation 1, groups 1, zero padding) on the implicit-GEMM engine."""

    def forward(self, x, act=ops.ACT_NONE, slope=0.0, in_act=None, out_pm=False, with_stats=False):
        """in_act / out_pm: the pre-masked gradient contract of ops.conv1d - only for a conv whose
        input comes straight from a fused conv + activation that has no other consumer (in_act, with
        out_pm on that producer)."""
        if self.dilation[0] != 1 or self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("m2d Conv1d: dilation / groups / non-zero padding are not on the hot path")
        return ops.conv1d(x, self.weight, self.bias, self.stride[0], self.padding[0], act, slope, in_act, out_pm,
                          with_stats)


    def forward_windows(self, track, T, hop, window, act=ops.ACT_NONE, slope=0.0, with_stats=False):
        """This (single input channel) conv over the windows of a padded track (B, S), read in place."""
        if self.in_channels != 1 or self.dilation[0] != 1 or self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("m2d Conv1d.forward_windows: single-channel, undilated convs only")
        return ops.conv1d_windows(track, T, hop, window, self.weight, self.bias, self.stride[0], self.padding[0],
                                  act, slope, with_stats)


class Linear(nn.Linear):
    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.linear(x, self.weight, self.bias, act, slope)


_BN_COUNT_BATCHED = [False]


class batched_bn_counters:
    """Inside: the training-mode BatchNorm layers of `module` do not launch one `num_batches_tracked += 1`
    each; all counters are advanced by ONE fused multi-tensor add on entry (a generator forward has 12 of
    them). Every BatchNorm1d of the module that is in training mode must run exactly once inside."""

    def __init__(self, module):
        self.counters = [m.num_batches_tracked for m in module.modules()
                         if isinstance(m, BatchNorm1d) and m.training and m.num_batches_tracked is not None]

    def __enter__(self):
        self.prev = _BN_COUNT_BATCHED[0]
        if self.counters and not self.prev:
            torch._foreach_add_(self.counters, 1)
            _BN_COUNT_BATCHED[0] = True
        return self

    def __exit__(self, *exc):
        _BN_COUNT_BATCHED[0] = self.prev
        return False


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d on (N, C) / (N, C, L) with the following ReLU / LeakyReLU and an
    optional residual add fused into the normalisation pass."""

    def forward(self, x, act=ops.ACT_NONE, slope=0.0, residual=None, sums=None, out=None):
        """sums: batch statistics of x from the producing conv's epilogue (Conv1d(..., with_stats=True)).
        out: ops.batch_norm's (no autograd graph only)."""
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError("m2d BatchNorm1d: only the reference's configuration is supported")
        if self.training and not _BN_COUNT_BATCHED[0]:
            self.num_batches_tracked.add_(1)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                              self.eps, self.momentum, act, slope, residual, sums if self.training else None, out)

    @torch.no_grad()
    def forward_then(self, x, sums, then, act=ops.ACT_NONE, slope=0.0, out=None, then_out=None):
        """No autograd graph, training mode, batch statistics handed over as `sums`: the normalisation with the pass
        that follows it fused in (kernels.bn_fwd_sums_pool / bn_fwd_sums_upsample2; the U-Net, phase3/archis/default.py).
        then = "pool":      -> (y, max_pool(y)); y into `out` (ops.batch_norm's)
        then = "upsample":  -> upsample2_linear(y) into `then_out`; y itself is not produced.
        Same values, same running-buffer update as forward() followed by the separate pass."""
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError("m2d BatchNorm1d: only the reference's configuration is supported")
        assert self.training and sums is not None and not torch.is_grad_enabled() and x.dim() == 3
        if not _BN_COUNT_BATCHED[0]:
            self.num_batches_tracked.add_(1)
        from . import kernels
        k = kernels.impl()
        x = x.contiguous()
        count = float(x.numel() // x.shape[1])
        args = (x, sums, count, self.weight, self.bias, self.running_mean, self.running_var, float(self.eps),
                float(self.momentum), int(act), float(slope))
        if then == "pool":
            return k.bn_fwd_sums_pool(*args, out=out)[:2]
        assert then == "upsample"
        return k.bn_fwd_sums_upsample2(*args, out=then_out)[0]

    @torch.no_grad()
    def observe(self, x):
        """Advance the running statistics with a batch without producing an output graph
        (the dead fc1 -> bn1 branch of LinearBlock, phase3/archis/default.py:184-187)."""
        if self.training:
            if not _BN_COUNT_BATCHED[0]:
                self.num_batches_tracked.add_(1)
            ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, True, self.eps,
                           self.momentum)


class GRU(nn.GRU):
    """nn.GRU(batch_first=True, unidirectional, no dropout): per layer one input-projection
    GEMM for all time steps, then the recurrent step kernels. Accepts a dense (B, T, in)
    tensor plus optional per-sequence lengths (the reference's packed input with equal
    lengths is bit-identical to the dense one)."""

    def forward(self, x, lengths=None):
        if not self.batch_first or self.bidirectional or self.dropout != 0.0 or not self.bias:
            raise NotImplementedError("m2d GRU: only batch_first / unidirectional / no-dropout is supported")
        params = []
        for layer in range(self.num_layers):
            params += [getattr(self, "weight_ih_l%d" % layer), getattr(self, "weight_hh_l%d" % layer),
                       getattr(self, "bias_ih_l%d" % layer), getattr(self, "bias_hh_l%d" % layer)]
        if self.num_layers <= 4:
            # every layer on the (layer, t) diagonal: T + L - 1 dependent launches instead of L * T
            return ops.gru_stack(x, params, lengths), None
        out = x
        for layer in range(self.num_layers):
            out = ops.gru_layer(out, *params[4 * layer:4 * layer + 4], lengths)
        return out, None


class WindowView:
    """(B, T, window) audio windows as a view of the padded track (B, S): window t of track b is
    track[b, t*hop : t*hop + window]. What utils.slice_audio_batch(..., lazy=True) stands for."""

    __slots__ = ("track", "T", "hop", "window")

    def __init__(self, track, T, hop, window):
        self.track, self.T, self.hop, self.window = track, int(T), int(hop), int(window)

    @staticmethod
    def of(x):
        """WindowView of an overlapping-window strided tensor (unfold of a padded track), else None."""
        if not torch.is_tensor(x) or x.dim() != 3 or x.stride(2) != 1 or not 0 < x.stride(1) < x.size(2):
            return None
        B, T, window = x.shape
        hop = x.stride(1)
        need = (T - 1) * hop + window
        S = x.stride(0) if B > 1 else need
        if S < need:
            return None
        # rows of `need` samples, S apart (the kernels take the row stride, not a dense tensor)
        return WindowView(torch.as_strided(x, (B, need), (S, 1), x.storage_offset()), T, hop, window)


class DrawTape:
    """The HOST random draws of a captured loop body (captured-graph mode of the small models, engine.Phase1Engine).
    The reference draws noise, interpolation weights and dropout masks from the default CPU generator inside the loop;
    a captured graph cannot. While a tape is installed (`draw_tape`) every `host_draw` hands out a STATIC device buffer
    instead - the first pass records (kind, shape, p) in program order, later passes (the capture) get the same buffers
    back in the same order - and `refill()` makes the same draws, in that order, on the host generator before each
    replay and copies them into the buffers: the generator is consumed exactly as the eager path consumes it."""

    def __init__(self):
        self.entries = []  # (kind, shape, p, buffer)
        self.cursor = 0

    def rewind(self):
        self.cursor = 0

    def draw(self, kind, shape, p, device, dtype):
        shape = tuple(shape)
        if self.cursor < len(self.entries):
            k, sh, pp, buf = self.entries[self.cursor]
            if (k, sh, pp) != (kind, shape, p):
                raise RuntimeError("DrawTape: the captured body drew %r where it had recorded %r" % ((kind, shape, p), (k, sh, pp)))
        else:
            buf = torch.zeros(shape, dtype=dtype, device=device)
            self.entries.append((kind, shape, p, buf))
        self.cursor += 1
        return buf

    def refill(self):
        for kind, shape, p, buf in self.entries:
            buf.copy_(to_device_async(_host_sample(kind, shape, p, buf.dtype), buf.device))


_TAPE = None


@contextlib.contextmanager
def draw_tape(tape):
    global _TAPE
    prev, _TAPE = _TAPE, tape
    try:
        yield tape
    finally:
        _TAPE = prev


def _host_sample(kind, shape, p, dtype):
    if kind == "randn":
        return torch.randn(*shape, dtype=dtype)
    if kind == "rand":
        return torch.rand(*shape, dtype=dtype)
    if kind == "bernoulli":
        return torch.empty(shape, dtype=dtype).bernoulli_(p)
    raise ValueError(kind)


def host_draw(kind, shape, device, p=0.0, dtype=torch.float32):
    """A draw from the default HOST generator ("randn" | "rand" | "bernoulli" with probability p), moved to `device`
    - or, under an installed DrawTape, the static buffer that stands for it."""
    if _TAPE is not None:
        return _TAPE.draw(kind, shape, float(p), torch.device(device), dtype)
    return to_device_async(_host_sample(kind, tuple(shape), float(p), dtype), device)


def lengths_tensor(lengths, T, device):
    """None when every sequence is full length (the training case), else an int32 tensor."""
    if lengths is None:
        return None
    ls = [int(v) for v in lengths]
    if all(v == T for v in ls):
        return None
    return torch.tensor(ls, dtype=torch.int32, device=device)


class Dropout(nn.Dropout):
    """nn.Dropout whose Bernoulli keep-mask can come from the HOST generator: the
    reference's CPU path draws `empty_like(x).bernoulli_(1-p)` from the default CPU RNG,
    and with `host_rng=True` (default) this layer consumes that stream identically, so
    seeded runs match the reference's CPU results. `host_rng=False` draws on the device."""

    host_rng = True

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.host_rng:
            keep = host_draw("bernoulli", x.shape, x.device, 1.0 - self.p, x.dtype)
        else:
            keep = torch.empty_like(x).bernoulli_(1.0 - self.p)
        return x * keep * (1.0 / (1.0 - self.p))


def head_activation(kind):
    """The reference's 'id' | 'relu' | 'tanh' switch -> (module kept for state/print parity,
    fused act code, needs_tanh)."""
    if kind == "id":
        return nn.Identity(), ops.ACT_NONE, False
    if kind == "relu":
        return nn.ReLU(True), ops.ACT_RELU, False
    if kind == "tanh":
        return nn.Tanh(), ops.ACT_NONE, True
    raise ValueError("unknown activation %r" % (kind,))
