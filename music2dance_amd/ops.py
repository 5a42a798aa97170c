"""Differentiable ops on top of the raw HIP kernels (kernels.py).

The gradient penalty differentiates the critic's input-gradient a second time
(losses.py:40-44 of the reference), so the critic ops are written as a set that is
closed under differentiation:

    conv1d(x, W)            backward = bwd_data(gy, W),  bwd_weight(x, gy)
    bwd_data(gy, W)         backward = conv1d(g, W) [d/d gy],  bwd_weight(g, gy) [d/d W]
    bwd_weight(x, gy)       backward = bwd_data(gy, g) [d/d x], conv1d(x, g) [d/d gy]

and the same triangle for linear layers (GEMM NT / NN / TN). Each backward is itself
built from these Functions, so `create_graph=True` records the second-order graph. The
fused activation (ReLU / LeakyReLU) is piecewise linear: its derivative is a mask that the
kernels apply while loading an operand or storing the result, so neither order of
derivative needs an extra elementwise pass over HBM.

Generator-only ops (BatchNorm, GRU, pooling, losses) are first-order.
"""
import contextlib
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import kernels

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def K():
    return kernels.impl()


# ---------------------------------------------------------------------------------------
# backward-pass pruning switches (results are unchanged; they only skip work whose result
# the caller discards). Process-global on purpose: they are set by the thread that calls
# backward() / autograd.grad() and read by autograd's device worker thread while that call blocks.
_state = {"inputs_only": False, "dead_inputs": set()}


@contextlib.contextmanager
def input_grads_only():
    """Inside: backward passes compute gradients w.r.t. op INPUTS only (weight and bias
    gradients return None). Used for the gradient penalty's first backward, which asks
    for d critic / d input only (losses.py:40-44); weights still receive their gradient
    through the double-backward graph."""
    prev = _state["inputs_only"]
    _state["inputs_only"] = True
    try:
        yield
    finally:
        _state["inputs_only"] = prev


@contextlib.contextmanager
def no_input_grad_for(*tensors):
    """Inside: ops whose input is one of `tensors` (matched by storage address) do not
    compute that input's gradient. The training engine uses it for the raw audio, whose
    gradient the reference computes and throws away."""
    alive = [t for t in tensors if t is not None]  # held for the whole scope: a live tensor's address is unique
    keys = {t.data_ptr() for t in alive}
    prev = _state["dead_inputs"]
    _state["dead_inputs"] = prev | keys
    try:
        yield
    finally:
        _state["dead_inputs"] = prev
        del alive


def _mask_of(act, slope, y):
    if act == ACT_NONE:
        return None, 0.0
    return y, (slope if act == ACT_LEAKY else 0.0)


def _c(t):
    return None if t is None else t.contiguous()


# --------------------------------------------------------------------------------------- conv1d
# Pre-masked gradients. y = act(conv(x)) needs h = gy * act'(y) in its backward. Read through a
# masked operand that costs the GEMM engine two loads per element in every tile that re-reads it
# (measured 10-30 % on the audio critic's layers). In a chain of fused conv + activation layers the
# consumer of y is the next conv, whose backward-data kernel can multiply its result by act'(its
# input) in the epilogue, once per element. Two flags, set by the MODULE that owns the chain
# (the op cannot see who consumes its output):
#   out_pm: every gradient that reaches this op's output is already multiplied by act'(y)
#           (all consumers of y are convs called with in_act);
#   in_act: (act, slope) of the op that produced x: every gradient this op returns for x - at any
#           order of differentiation - is multiplied by act'(x) in the producing kernel's epilogue.
# Activations are piecewise linear, so masks carry no gradient, and for every node below the rule
# is the same at first and second order: gradients w.r.t. `x` get the in_act epilogue, gradients
# w.r.t. `gy` / `h` get the act'(y) epilogue (that one is plain calculus: h = gy * act'(y)), and an
# operand is read through a mask only when nobody pre-multiplied it.
_FUSED_BIAS = os.environ.get("M2D_FUSED_BIAS", "1") != "0"  # dev switch for A/B timing
_PREMASK = os.environ.get("M2D_PREMASK", "1") != "0"  # dev switch for A/B timing: 0 = masked operand loads


def _slope_of(act, slope):
    return slope if act == ACT_LEAKY else 0.0


class _Conv1dAct(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope, in_act, out_pm, with_stats):
        ctx.set_materialize_grads(False)
        x, w, b = _c(x), _c(w), _c(b)
        sums = None
        if with_stats:
            y, sums = K().conv1d_fwd(x, w, b, stride, pad, act, slope, with_stats=True)
            ctx.mark_non_differentiable(sums)
        else:
            y = K().conv1d_fwd(x, w, b, stride, pad, act, slope)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.cfg = (stride, pad, act, slope, b is not None, x.data_ptr(), in_act, bool(out_pm) and act != ACT_NONE)
        return y, sums

    @staticmethod
    def backward(ctx, gy, _g_sums=None):
        if gy is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 10
        x, w, y = ctx.saved_tensors
        stride, pad, act, slope, has_bias, xkey, in_act, out_pm = ctx.cfg
        ymask, yslope = _mask_of(act, slope, y)
        hmask = None if out_pm else ymask  # operand mask only when the gradient is not pre-multiplied
        xmask, xslope = (x, _slope_of(*in_act)) if in_act else (None, 0.0)
        gy = _c(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and xkey not in _state["dead_inputs"]:
            gx = _Conv1dBwdData.apply(gy, w, hmask, ymask, xmask, x.shape[2], stride, pad, yslope, xslope)
        if not _state["inputs_only"]:
            want_b = has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                # the bias gradient rides in the weight-gradient launch (an all-ones operand column)
                gw, gb = _Conv1dBwdWeight.apply(x, gy, hmask, ymask, xmask, w.shape[2], stride, pad, yslope,
                                                xslope, want_b and _FUSED_BIAS)
                if want_b and not _FUSED_BIAS:
                    gb = _ChannelSum.apply(gy, hmask, ymask, yslope)
            elif want_b:
                gb = _ChannelSum.apply(gy, hmask, ymask, yslope)
        return gx, gw, gb, None, None, None, None, None, None, None


class _Conv1dBwdData(Function):
    """dx = act_in'(x) * conv_transpose(h, W), h = gy * act'(y) (read through `hmask`, or already
    multiplied when hmask is None). Differentiable w.r.t. gy and W."""

    @staticmethod
    def forward(ctx, gy, w, hmask, ymask, xmask, L, stride, pad, yslope, xslope):
        ctx.set_materialize_grads(False)
        gy, w = _c(gy), _c(w)
        dx = K().conv1d_bwd_data(gy, w, L, stride, pad, hmask, yslope, xmask, xslope)
        ctx.save_for_backward(gy, w, hmask, ymask)
        ctx.cfg = (stride, pad, yslope)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        # g arrives multiplied by act_in'(x) whenever xmask was applied: its producers are the
        # second-order nodes of the layer that made x, which all end in that layer's act' epilogue
        if g is None:
            return (None,) * 10
        gy, w, hmask, ymask = ctx.saved_tensors
        stride, pad, yslope = ctx.cfg
        g = _c(g)
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = K().conv1d_fwd(g, w, None, stride, pad, 0, 0.0, None, ymask, yslope)
        if ctx.needs_input_grad[1]:
            g_w = K().conv1d_bwd_weight(g, gy, w.shape[2], stride, pad, hmask, yslope)
        return (g_gy, g_w) + (None,) * 8


class _Conv1dBwdWeight(Function):
    """dW = correlate(x, h) and, with `with_bias`, db = sum h; h = gy * act'(y). Differentiable
    w.r.t. x and gy."""

    @staticmethod
    def forward(ctx, x, gy, hmask, ymask, xmask, ks, stride, pad, yslope, xslope, with_bias):
        ctx.set_materialize_grads(False)
        x, gy = _c(x), _c(gy)
        ctx.save_for_backward(x, gy, hmask, ymask, xmask)
        ctx.cfg = (stride, pad, yslope, xslope)
        if with_bias:
            return K().conv1d_bwd_weight(x, gy, ks, stride, pad, hmask, yslope, with_bias=True)
        return K().conv1d_bwd_weight(x, gy, ks, stride, pad, hmask, yslope), None

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g_b):
        if g is None and g_b is None:
            return (None,) * 11
        x, gy, hmask, ymask, xmask = ctx.saved_tensors
        stride, pad, yslope, xslope = ctx.cfg
        g_x = g_gy = None
        if g is not None:
            g = _c(g)
            if ctx.needs_input_grad[0]:
                g_x = K().conv1d_bwd_data(gy, g, x.shape[2], stride, pad, hmask, yslope, xmask, xslope)
            if ctx.needs_input_grad[1]:
                g_gy = K().conv1d_fwd(x, g, None, stride, pad, 0, 0.0, None, ymask, yslope)
        if g_b is not None and ctx.needs_input_grad[1]:
            e = g_b.view(1, -1, 1).expand(gy.shape)
            if ymask is not None:
                e = e * torch.where(ymask > 0, torch.ones_like(ymask), torch.full_like(ymask, yslope))
            g_gy = e.contiguous() if g_gy is None else g_gy + e
        return (g_x, g_gy) + (None,) * 9


class _ChannelSum(Function):
    """sum over (batch, length) of h = gy * act'(y): bias gradients (`hmask` None: gy is already h)."""

    @staticmethod
    def forward(ctx, gy, hmask, ymask, mslope):
        ctx.set_materialize_grads(False)
        gy = _c(gy)
        ctx.save_for_backward(ymask)
        ctx.mslope = mslope
        ctx.shape = gy.shape
        return K().channel_sums(gy, hmask, mslope)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:
            return (None,) * 4
        (ymask,) = ctx.saved_tensors
        shape = ctx.shape
        view = (1, -1, 1) if len(shape) == 3 else (1, -1)
        out = g.view(view).expand(shape)
        if ymask is not None:
            out = out * torch.where(ymask > 0, torch.ones_like(ymask), torch.full_like(ymask, ctx.mslope))
        return out.contiguous(), None, None, None


def conv1d(x, weight, bias=None, stride=1, padding=0, act=ACT_NONE, slope=0.0, in_act=None, out_pm=False,
           with_stats=False):
    """nn.Conv1d forward with optional fused ReLU / LeakyReLU, twice differentiable.
    in_act / out_pm: the pre-masked gradient contract described above (module-level promise).
    with_stats: return (y, sums) - the per-channel sum / sum of squares of y (float64, 2*Cout) taken in the
    conv's epilogue, for a BatchNorm that follows (batch_norm(..., sums=sums))."""
    if not _PREMASK:
        in_act, out_pm = None, False
    if in_act is not None:
        in_act = (int(in_act[0]), float(in_act[1]))
        if in_act[0] == ACT_NONE:
            in_act = None
    y, sums = _Conv1dAct.apply(x, weight, bias, int(stride), int(padding), int(act), float(slope), in_act,
                               bool(out_pm), bool(with_stats))
    return (y, sums) if with_stats else y


class _Conv1dWindows(Function):
    """Conv1d(1, Cout, ...) (+ fused activation) over the windows of a padded track, read in place
    (kernels.conv1d_fwd_windows). First-order only (generator path); the track gets no gradient."""

    @staticmethod
    def forward(ctx, track, w, b, T, hop, window, stride, pad, act, slope, with_stats):
        ctx.set_materialize_grads(False)
        w, b = _c(w), _c(b)
        sums = None
        if with_stats:
            y, sums = K().conv1d_fwd_windows(track, T, hop, window, w, b, stride, pad, act, slope, with_stats=True)
            ctx.mark_non_differentiable(sums)
        else:
            y = K().conv1d_fwd_windows(track, T, hop, window, w, b, stride, pad, act, slope)
        ctx.save_for_backward(track, w, y if act else None)
        ctx.cfg = (T, hop, window, stride, pad, act, slope, b is not None)
        return y, sums

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _g_sums=None):
        if gy is None:
            return (None,) * 11
        track, w, y = ctx.saved_tensors
        T, hop, window, stride, pad, act, slope, has_bias = ctx.cfg
        mask, mslope = _mask_of(act, slope, y)
        gy = _c(gy)
        gw = gb = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            want_b = has_bias and ctx.needs_input_grad[2]
            out = K().conv1d_bwd_weight_windows(track, T, hop, window, gy, w.shape[2], stride, pad, mask, mslope,
                                                with_bias=want_b)
            gw, gb = out if want_b else (out, None)
        return (None, gw, gb) + (None,) * 8


def conv1d_windows(track, T, hop, window, weight, bias=None, stride=1, padding=0, act=ACT_NONE, slope=0.0,
                   with_stats=False):
    """conv1d over the (B*T, 1, window) windows of `track` (B, S) without materialising them."""
    if track.requires_grad:
        raise NotImplementedError("conv1d_windows: no gradient w.r.t. the audio track (materialise the slices)")
    y, sums = _Conv1dWindows.apply(track, weight, bias, int(T), int(hop), int(window), int(stride), int(padding),
                                   int(act), float(slope), bool(with_stats))
    return (y, sums) if with_stats else y


# --------------------------------------------------------------------------------------- tanh heads
class _Tanh(Function):
    """nn.Tanh of the `activ: tanh` heads, twice differentiable through _TanhBwd (the critics' heads sit under the
    gradient penalty's double backward)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        y = K().tanh_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None
        (y,) = ctx.saved_tensors
        return _TanhBwd.apply(gy, y)


class _TanhBwd(Function):
    """gx = gy * (1 - y^2); differentiable w.r.t. gy (the same map applied to g) and y (-2 y g gy)."""

    @staticmethod
    def forward(ctx, gy, y):
        ctx.set_materialize_grads(False)
        gy = _c(gy)
        ctx.save_for_backward(gy, y)
        return K().tanh_bwd(gy, y)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None
        gy, y = ctx.saved_tensors
        g = _c(g)
        g_gy = K().tanh_bwd(g, y) if ctx.needs_input_grad[0] else None
        g_y = K().tanh_bwd_bwd(g, gy, y) if ctx.needs_input_grad[1] else None
        return g_gy, g_y


def tanh(x):
    return _Tanh.apply(x)


# --------------------------------------------------------------------------------------- linear
class _LinearAct(Function):
    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        ctx.set_materialize_grads(False)
        x, w, b = _c(x), _c(w), _c(b)
        y = K().gemm(0, x, w, b, act, slope)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.cfg = (act, slope, b is not None, x.data_ptr())
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 5
        x, w, y = ctx.saved_tensors
        act, slope, has_bias, xkey = ctx.cfg
        mask, mslope = _mask_of(act, slope, y)
        gy = _c(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0] and xkey not in _state["dead_inputs"]:
            gx = _LinearBwdData.apply(gy, w, mask, mslope)
        if not _state["inputs_only"]:
            if ctx.needs_input_grad[1]:
                gw = _LinearBwdWeight.apply(gy, x, mask, mslope)
            if has_bias and ctx.needs_input_grad[2]:
                gb = _ChannelSum.apply(gy, mask, mask, mslope)
        return gx, gw, gb, None, None


class _LinearBwdData(Function):
    """dx = (gy * act'(y)) W."""

    @staticmethod
    def forward(ctx, gy, w, mask, mslope):
        ctx.set_materialize_grads(False)
        gy, w = _c(gy), _c(w)
        ctx.save_for_backward(gy, w, mask)
        ctx.mslope = mslope
        return K().gemm(1, gy, w, a_mask=mask, a_mask_slope=mslope)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 4
        gy, w, mask = ctx.saved_tensors
        g = _c(g)
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = K().gemm(0, g, w, out_mask=mask, out_mask_slope=ctx.mslope)
        if ctx.needs_input_grad[1]:
            g_w = K().gemm(2, gy, g, a_mask=mask, a_mask_slope=ctx.mslope)
        return g_gy, g_w, None, None


class _LinearBwdWeight(Function):
    """dW = (gy * act'(y))^T x."""

    @staticmethod
    def forward(ctx, gy, x, mask, mslope):
        ctx.set_materialize_grads(False)
        gy, x = _c(gy), _c(x)
        ctx.save_for_backward(gy, x, mask)
        ctx.mslope = mslope
        return K().gemm(2, gy, x, a_mask=mask, a_mask_slope=mslope)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 4
        gy, x, mask = ctx.saved_tensors
        g = _c(g)
        g_gy = g_x = None
        if ctx.needs_input_grad[0]:
            g_gy = K().gemm(0, x, g, out_mask=mask, out_mask_slope=ctx.mslope)
        if ctx.needs_input_grad[1]:
            g_x = K().gemm(1, gy, g, a_mask=mask, a_mask_slope=ctx.mslope)
        return g_gy, g_x, None, None


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    """nn.Linear forward (x: (N, in)) with optional fused activation, twice differentiable."""
    if x.dim() != 2:
        lead = x.shape[:-1]
        return linear(x.reshape(-1, x.shape[-1]), weight, bias, act, slope).view(*lead, weight.shape[0])
    return _LinearAct.apply(x, weight, bias, int(act), float(slope))


# --------------------------------------------------------------------------------------- batch norm
# Synchronised BatchNorm (optional): with a process group set here the batch
# statistics (and the two backward sums) are all-reduced across the data-parallel ranks, so the
# generator normalises over the GLOBAL batch exactly like a single process would. Off = per-rank
# statistics (the torch-DDP default). The exchange is 2*C doubles per BatchNorm layer and pass.
_sync_bn = {"group": None, "on": False}


def set_sync_batchnorm(on=True, group=None):
    import torch.distributed as dist
    _sync_bn["on"] = bool(on) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    _sync_bn["group"] = group
    return _sync_bn["on"]


def sync_batchnorm_active():
    return bool(_sync_bn["on"])


def _sync_world():
    if not _sync_bn["on"]:
        return 1
    import torch.distributed as dist
    return dist.get_world_size(_sync_bn["group"])


def _all_reduce_sums(sums):
    import torch.distributed as dist
    out = sums.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=_sync_bn["group"])
    return out


_sync_counts = {}


def _global_count(local_count, device):
    """Elements per channel over ALL ranks. Reduced once per distinct local count (a host read, i.e. a device sync: not
    something to do per layer and step) and cached - which is only sound when every rank sees the same sequence of
    local counts (a rank with a cached value skips the collective another rank would wait in for ever). So EQUAL shards
    are required and checked: the first reduction of a count also reduces its minimum and maximum over the ranks
    (as MAX of (-count, count), the same collective on every rank), and EVERY rank raises when they differ - a test on
    the sum alone lets a rank whose count happens to equal the mean (10 / 12 / 14) pass, cache and walk into the next
    collective alone (use drop_last / equal per-rank batches with synchronised BatchNorm)."""
    import torch.distributed as dist
    key = (float(local_count), id(_sync_bn["group"]))
    g = _sync_counts.get(key)
    if g is None:
        t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
        mm = torch.tensor([-float(local_count), float(local_count)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_sync_bn["group"])
        dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=_sync_bn["group"])
        g = float(t.item())
        lo, hi = -float(mm[0].item()), float(mm[1].item())
        world = dist.get_world_size(_sync_bn["group"])
        if hi - lo > 0.5:
            raise RuntimeError("synchronised BatchNorm needs equal shards on every rank: this rank has %d elements per "
                               "channel, the %d ranks hold between %d and %d (%d together)"
                               % (int(local_count), world, int(lo), int(hi), int(g)))
        _sync_counts[key] = g
    return g


def _bn_forward(x, gamma, beta, running_mean, running_var, residual, training, eps, momentum, act, slope, sums, out=None):
    """-> (y, mean, invstd, world); out: see kernels.bn_fwd (the result written into a channel block of a wider buffer)"""
    x, residual = _c(x), _c(residual)
    world = 1
    if training:
        world = _sync_world()
    kw = {} if out is None else {"out": out}
    if training and sums is None and world == 1:
        # one process, no sums handed over: statistics, mean / invstd and the running buffers in one launch
        y, mean, invstd = K().bn_fwd(x, gamma, beta, running_mean, running_var, True, eps, momentum, act, slope,
                                     residual, **kw)
    elif training:
        # batch statistics as raw sums: from the producing conv's epilogue when it supplied them
        if sums is None:
            sums = K().bn_stats(x)
        count = float(x.numel() // x.shape[1])
        if world > 1:
            sums = _all_reduce_sums(sums)
            count = _global_count(count, x.device)
        y, mean, invstd = K().bn_fwd_sums(x, sums, count, gamma, beta, running_mean, running_var, eps, momentum,
                                          act, slope, residual, **kw)
    else:
        y, mean, invstd = K().bn_fwd(x, gamma, beta, running_mean, running_var, False, eps, momentum, act,
                                     slope, residual, **kw)
    return x, y, mean, invstd, world


class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, eps, momentum, act, slope, sums):
        ctx.set_materialize_grads(False)
        x, y, mean, invstd, world = _bn_forward(x, gamma, beta, running_mean, running_var, residual, training, eps,
                                                momentum, act, slope, sums)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.cfg = (training, act, slope, residual is not None, world)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if gy is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 12
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        training, act, slope, has_res, world = ctx.cfg
        if not training:
            raise NotImplementedError("m2d BatchNorm: backward in eval mode is not part of the training path")
        gy = _c(gy)
        if world > 1:
            local = K().bn_bwd_stats(gy, x, gamma, beta, mean, invstd, act, slope)
            glob = _all_reduce_sums(local)
            dx, dgamma, dbeta = K().bn_bwd_sums(gy, x, gamma, beta, mean, invstd, local, glob,
                                                _global_count(x.numel() // x.shape[1], x.device), act, slope)
        else:
            dx, dgamma, dbeta = K().bn_bwd(gy, x, gamma, beta, mean, invstd, act, slope)
        return dx, dgamma, dbeta, None, None, (gy if has_res else None), None, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, act=ACT_NONE,
               slope=0.0, residual=None, sums=None, out=None):
    """y = residual + act(batch_norm(x)); running buffers are updated in place when training.
    sums: the batch statistics of x as raw sums (conv1d(..., with_stats=True)), else computed here.
    out (no autograd graph only): a (B, C, L) view whose samples are out.stride(0) apart - the result is written there
    (a channel block of a wider buffer: UBlock's skip concatenations made in place)."""
    if out is not None:
        assert not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, gamma, beta, residual)))
        return _bn_forward(x, gamma, beta, running_mean, running_var, residual, bool(training), float(eps), float(momentum),
                           int(act), float(slope), sums, out)[1]
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, residual, bool(training), float(eps),
                               float(momentum), int(act), float(slope), sums)


# --------------------------------------------------------------------------------------- GRU
class _GRULayer(Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lengths, save):
        ctx.set_materialize_grads(False)
        x, w_ih, w_hh = _c(x), _c(w_ih), _c(w_hh)
        B, T, I = x.shape
        H = w_hh.shape[1]
        k = K()
        gi = k.gemm(0, x.view(B * T, I), w_ih, b_ih).view(B, T, 3 * H)
        out, saved = k.gru_layer_fwd(gi, k.transposed(w_hh), b_hh, lengths, save)
        if save:
            ctx.save_for_backward(x, w_ih, w_hh, out, saved, lengths)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        if dout is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 7
        x, w_ih, w_hh, out, saved, lengths = ctx.saved_tensors
        B, T, I = x.shape
        H = w_hh.shape[1]
        k = K()
        dgi, dgh = k.gru_layer_bwd(_c(dout), out, saved, w_hh, lengths)
        dgi2, dgh2 = dgi.view(B * T, 3 * H), dgh.view(B * T, 3 * H)
        dx = k.gemm(1, dgi2, w_ih).view(B, T, I) if ctx.needs_input_grad[0] else None
        dw_ih = k.gemm(2, dgi2, x.view(B * T, I))
        hprev = torch.cat((out.new_zeros(B, 1, H), out[:, :-1]), 1).contiguous().view(B * T, H)
        dw_hh = k.gemm(2, dgh2, hprev)
        db_ih = k.channel_sums(dgi2.view(B * T, 3 * H, 1))
        db_hh = k.channel_sums(dgh2.view(B * T, 3 * H, 1))
        return dx, dw_ih, dw_hh, db_ih, db_hh, None, None


class _GRUStack(Function):
    """All layers of an nn.GRU at once on the (layer, t) diagonal (T + L - 1 step launches)."""

    @staticmethod
    def forward(ctx, x, lengths, save, *params):
        ctx.set_materialize_grads(False)
        L = len(params) // 4
        w_ih = [_c(params[4 * l]) for l in range(L)]
        w_hh = [_c(params[4 * l + 1]) for l in range(L)]
        b_ih = [_c(params[4 * l + 2]) for l in range(L)]
        b_hh = [_c(params[4 * l + 3]) for l in range(L)]
        x = _c(x)
        B, T, I = x.shape
        H = w_hh[0].shape[1]
        k = K()
        gi0 = k.gemm(0, x.view(B * T, I), w_ih[0], b_ih[0]).view(B, T, 3 * H)
        outs, saved = k.gru_stack_fwd(gi0, [None] + [k.transposed(w) for w in w_ih[1:]], [None] + b_ih[1:],
                                      [k.transposed(w) for w in w_hh], b_hh, lengths, save)
        if save:
            ctx.save_for_backward(x, lengths, *w_ih, *w_hh, *outs, *saved)
            ctx.L = L
        return outs[-1]

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        if dout is None:
            return (None,) * (3 + 4 * ctx.L)
        L = ctx.L
        t = ctx.saved_tensors
        x, lengths = t[0], t[1]
        w_ih, w_hh = list(t[2:2 + L]), list(t[2 + L:2 + 2 * L])
        outs, saved = list(t[2 + 2 * L:2 + 3 * L]), list(t[2 + 3 * L:2 + 4 * L])
        B, T, I = x.shape
        H = w_hh[0].shape[1]
        k = K()
        dgi, dgh = k.gru_stack_bwd(_c(dout), outs, saved, w_hh, [None] + w_ih[1:], lengths,
                                   persistent=os.environ.get("M2D_GRU_BWD_PERSIST", "1") != "0")
        grads = []
        for l in range(L):
            dgi2, dgh2 = dgi[l].view(B * T, 3 * H), dgh[l].view(B * T, 3 * H)
            inp = x.view(B * T, I) if l == 0 else outs[l - 1].view(B * T, H)
            hprev = torch.cat((outs[l].new_zeros(B, 1, H), outs[l][:, :-1]), 1).contiguous().view(B * T, H)
            grads += [k.gemm(2, dgi2, inp), k.gemm(2, dgh2, hprev), k.channel_sums(dgi2.view(B * T, 3 * H, 1)),
                      k.channel_sums(dgh2.view(B * T, 3 * H, 1))]
        dx = k.gemm(1, dgi[0].view(B * T, 3 * H), w_ih[0]).view(B, T, I) if ctx.needs_input_grad[0] else None
        return (dx, None, None) + tuple(grads)


def gru_stack(x, params, lengths=None):
    """nn.GRU(batch_first, h0 = 0) with len(params) // 4 layers; params = [w_ih, w_hh, b_ih, b_hh] per layer."""
    save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    return _GRUStack.apply(x, lengths, save, *params)


def gru_layer(x, w_ih, w_hh, b_ih, b_hh, lengths=None):
    """One nn.GRU layer (batch_first, h0 = 0) over a whole (B, T, in) sequence."""
    save = torch.is_grad_enabled() and any(t.requires_grad for t in (x, w_ih, w_hh, b_ih, b_hh))
    return _GRULayer.apply(x, w_ih, w_hh, b_ih, b_hh, lengths, save)


# --------------------------------------------------------------------------------------- GP
def gp_interpolate(real2d, fake2d, alpha):
    """alpha*real + (1-alpha)*fake on detached inputs (losses.py:20); alpha: (B,)."""
    return K().gp_interpolate(_c(real2d.detach()), _c(fake2d.detach()), _c(alpha))


class _GPPenalty(Function):
    @staticmethod
    def forward(ctx, g, lp):
        ctx.set_materialize_grads(False)
        g = _c(g)
        pen, norms = K().gp_penalty_fwd(g, lp)
        ctx.save_for_backward(g, norms)
        ctx.lp = lp
        return pen

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if gout is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 2
        g, norms = ctx.saved_tensors
        return K().gp_penalty_bwd(g, norms, _c(gout), ctx.lp), None


def gp_penalty(grad2d, lp=False):
    """mean_b (||g_b||_2 - 1)^2 (GP, eps 1e-12 inside the sqrt) or mean_b max(0, ||g_b|| - 1)^2 (LP)."""
    return _GPPenalty.apply(grad2d, bool(lp))


# --------------------------------------------------------------------------------------- losses
class _L1Mean(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        return K().l1_mean_fwd(a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if gout is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 2
        a, b = ctx.saved_tensors
        gout = _c(gout)
        ga = K().l1_mean_bwd(a, b, gout) if ctx.needs_input_grad[0] else None
        gb = -K().l1_mean_bwd(a, b, gout) if ctx.needs_input_grad[1] else None
        return ga, gb


def l1_mean(a, b):
    """torch.nn.L1Loss(reduction='mean')(a, b); both tensors must share one memory layout."""
    return _L1Mean.apply(a, b)


class _TVMean(Function):
    @staticmethod
    def forward(ctx, store, B, C, T, sb, sc, st):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(store)
        ctx.cfg = (B, C, T, sb, sc, st)
        return K().tv_mean_fwd(store, B, C, T, sb, sc, st)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        if gout is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 7
        (store,) = ctx.saved_tensors
        return (K().tv_mean_bwd(store, _c(gout), *ctx.cfg),) + (None,) * 6


def tv_mean(seq):
    """mean |x[:, :, 1:] - x[:, :, :-1]| of a (B, C, T) tensor or permuted view of a dense one."""
    B, C, T = seq.shape
    if seq.is_contiguous():
        return _TVMean.apply(seq, B, C, T, C * T, T, 1)
    base = seq.permute(0, 2, 1)
    if base.is_contiguous():  # (B, T, C) storage viewed as (B, C, T): the generator's native layout
        return _TVMean.apply(base, B, C, T, T * C, 1, C)
    return tv_mean(seq.contiguous())


def jerk_mean(seq):
    """losses.jerkiness of a (B, C, T) tensor (or permuted view of a dense (B, T, C) one): a metric, no gradient."""
    B, C, T = seq.shape
    seq = seq.detach()
    if seq.dtype != torch.float32:
        seq = seq.float()
    if seq.is_contiguous():
        return K().jerk_mean_fwd(seq, B, C, T, C * T, T, 1)
    base = seq.permute(0, 2, 1)
    if base.is_contiguous():
        return K().jerk_mean_fwd(base, B, C, T, T * C, 1, C)
    return jerk_mean(seq.contiguous())


# --------------------------------------------------------------------------------------- U-Net resampling
class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        x = _c(x)
        ctx.save_for_backward(x)
        return K().maxpool2_fwd(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if gy is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 1
        (x,) = ctx.saved_tensors
        return K().maxpool2_bwd(x, _c(gy))


class _Upsample2(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return K().upsample2_fwd(_c(x))

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if gy is None:  # no gradient reaches this node (e.g. the penalty pass's forward graph)
            return (None,) * 1
        return K().upsample2_bwd(_c(gy))


def maxpool2(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return K().maxpool2_fwd(x)   # (takes a channel block of a wider buffer as it is: UBlock's in-place concatenations)
    return _MaxPool2.apply(x)


def upsample2_linear(x, out=None):
    """nn.Upsample(scale_factor=2, mode='linear', align_corners=False) on (B, C, L). out (no autograd graph only): as
    batch_norm's - a channel block of a wider buffer to write into."""
    if out is not None:
        assert not (torch.is_grad_enabled() and x.requires_grad)
        return K().upsample2_fwd(_c(x), out=out)
    return _Upsample2.apply(x)
