"""Tensor-level wrappers over the C-ABI (include/m2d.h): raw kernels, no autograd.

PyTorch is plumbing here: it owns device memory (outputs and scratch come from the
caching allocator, so the calls are hipGraph-capture safe) and the current HIP stream.
Every method enqueues hand-written gfx950 kernels and nothing else; inputs must be
contiguous fp32 tensors on a HIP device, otherwise the call raises (no fallback).
"""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.M2dError("m2d kernels need HIP device tensors (got %s); there is no CPU path" % t.device)
        if t.dtype != torch.float32:
            raise _lib.M2dError("m2d kernels are fp32 only (got %s)" % t.dtype)
        if not t.is_contiguous():
            raise _lib.M2dError("m2d kernels need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.M2dError("tensors on different devices")
    return dev


def _chk_track(track, dev, T, hop, window):
    """(B, >= (T-1)*hop + window) fp32 rows on `dev`, unit stride inside a row; -> (B, row stride)."""
    if not track.is_cuda or track.device != dev or track.dtype != torch.float32 or track.dim() != 2:
        raise _lib.M2dError("window view: the track must be a 2-D fp32 tensor on %s" % dev)
    if track.stride(1) != 1 or track.size(1) < (T - 1) * hop + window:
        raise _lib.M2dError("window view: rows must be dense and hold (T-1)*hop + window samples")
    B = track.size(0)
    S = track.stride(0) if B > 1 else track.size(1)
    if S < (T - 1) * hop + window:
        raise _lib.M2dError("window view: overlapping tracks")
    return B, S


_STREAM_SCRATCH = {}
_FUSED_SPLITK = os.environ.get("M2D_FUSED_SPLITK", "1") != "0"
_TICKET_WORDS = 16384   # 4 bytes per output tile of a split-K launch (include/m2d.h: 64 KB covers every shape)
_BN_MAX_C = 4096        # widest channel / feature count a reducing call may have (the reference's widest: 1024)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _stream(dev):
    # raw hipStream_t of the current stream without building a torch.cuda.Stream object
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = torch._C._cuda_getCurrentRawStream(idx)
    if _PRIVATE is not None:
        _PRIVATE.adopt((idx, s))
    elif _FUSED_SPLITK and (idx, s) not in _STREAM_SCRATCH and not _capturing():
        # first launch on this stream: its zero-kept ticket scratch (include/m2d.h: m2d_stream_scratch_set) - split-K
        # launches then finish without the second (reduction) launch. Never created inside a capture (it would live in
        # that graph's private pool behind a captured memset): a capture without `private_scratch` keeps the
        # two-launch split-K, which needs no state.
        t = torch.zeros(_TICKET_WORDS, dtype=torch.int32, device=dev)
        _STREAM_SCRATCH[(idx, s)] = t
        _lib.check(_lib.lib().m2d_stream_scratch_set(s, t.data_ptr(), t.numel() * 4), "m2d_stream_scratch_set")
    return s


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev):
    """Device guard for a launch: a no-op when `dev` is already current (the common case)."""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(dev)


_WS_BYTES = {}


def _ws_bytes(name, *args):
    """Workspace size queries are pure functions of the shape: ask the library once per shape."""
    if "M2D_PLAN" in os.environ:  # tuning builds: the plan (and its workspace) follows the environment
        return getattr(_lib.lib(), name)(*args)
    key = (name,) + args
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = getattr(_lib.lib(), name)(*args)
    return n


def _ws(nbytes, dev):
    if nbytes <= 0:
        return None
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)


_BN_SCRATCH = {}


def _bn_scratch(dev, C):
    """The zero-kept accumulator scratch of the reducing BatchNorm / channel-sum calls (include/m2d.h:
    m2d_bn_scratch_bytes): one per (device, stream), of FIXED size (captured graphs keep its address: it is never
    re-allocated), zeroed once here and left zeroed by every call. None inside a capture that has not been given one
    (`private_scratch`): the call then takes the stateless three-launch form."""
    if C > _BN_MAX_C:
        return None  # wider than the fixed scratch (a GRU with hidden size > _BN_MAX_C / 3): the stateless three-launch form
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    t = _BN_SCRATCH.get(key)
    if t is None:
        if _capturing() or _PRIVATE is not None:
            return None
        t = _BN_SCRATCH[key] = torch.zeros((2 * _BN_MAX_C + 8,), dtype=torch.float64, device=dev)
    return t


_PRIVATE = None   # the private_scratch scope being captured under, if any


class private_scratch:
    """Scope for CAPTURING a graph: launches made inside it - on the capture stream and on every stream forked from it -
    take their zero-kept scratch (split-K tickets, BatchNorm / channel-sum accumulators) from buffers owned by this
    object instead of the streams' own. A captured launch bakes the scratch ADDRESS in; two graphs captured on one
    stream (torch captures every graph on the same class-level stream by default) would otherwise share it, and
    replaying them concurrently - the phase-2 generator-forward graph runs on the generator stream under the critic
    graph of the previous body - has both sets of kernels counting arrivals in the same words. Keep the object alive
    as long as the graph. Everything is allocated in __init__, i.e. outside the capture: `nstreams` sets (the capture
    stream + the side streams the modules fork); a further stream gets none and its calls take the stateless forms."""

    def __init__(self, dev, nstreams=4):
        self.dev = dev
        self.free = [(torch.zeros(_TICKET_WORDS, dtype=torch.int32, device=dev) if _FUSED_SPLITK else None,
                      torch.zeros((2 * _BN_MAX_C + 8,), dtype=torch.float64, device=dev)) for _ in range(nstreams)]
        self.by_stream = {}
        self.saved = {}

    @staticmethod
    def _register(raw, t):
        if t is None:
            _lib.check(_lib.lib().m2d_stream_scratch_set(raw, None, 0), "m2d_stream_scratch_set")
        else:
            _lib.check(_lib.lib().m2d_stream_scratch_set(raw, t.data_ptr(), t.numel() * 4), "m2d_stream_scratch_set")

    def adopt(self, key):
        """First launch on stream `key` = (device index, raw stream) inside the scope: give it one of the sets."""
        if key in self.by_stream:
            return
        pair = self.free.pop() if self.free else (None, None)
        self.by_stream[key] = pair
        self.saved[key] = (_STREAM_SCRATCH.get(key), _BN_SCRATCH.get(key))
        for table, t in ((_STREAM_SCRATCH, pair[0]), (_BN_SCRATCH, pair[1])):
            if t is None:
                table.pop(key, None)
            else:
                table[key] = t
        if _FUSED_SPLITK:
            self._register(key[1], pair[0])

    def __enter__(self):
        global _PRIVATE
        self._outer, _PRIVATE = _PRIVATE, self
        return self

    def __exit__(self, *exc):
        global _PRIVATE
        _PRIVATE = self._outer
        for key, (old_t, old_b) in self.saved.items():
            for table, t in ((_STREAM_SCRATCH, old_t), (_BN_SCRATCH, old_b)):
                if t is None:
                    table.pop(key, None)
                else:
                    table[key] = t
            if _FUSED_SPLITK:
                self._register(key[1], old_t)
        self.saved = {}
        return False


def set_plan_model(model):
    """Which cost model ranks the GEMM engine's launch plans (include/m2d.h: m2d_plan_model_set): 4 where the streams of
    a loop body overlap (two-branch phase-3 critic), 5 where launches run one after the other (phase 2, pose-only critic).
    PROCESS-WIDE, and plans decide summation orders: a program sets it once, before its first launch (bench.py does for
    its c2 / c5 presets, the phase-2 train script does) - the engines do not touch it (round 5: an engine that switched it
    in its constructor changed the arithmetic of whatever ran next in the process). An explicit M2D_PLAN_MODEL in the
    environment wins. Cached workspace sizes follow the plans: dropped."""
    if "M2D_PLAN_MODEL" in os.environ or not hasattr(_lib.lib(), "m2d_plan_model_set"):
        return
    if _lib.lib().m2d_plan_model_get() != int(model):
        _lib.check(_lib.lib().m2d_plan_model_set(int(model)), "m2d_plan_model_set")
        _WS_BYTES.clear()


def reset_scratch():
    """Re-zero every zero-kept scratch buffer of the streams' own (not the graphs' private ones). A launch that was
    aborted mid-way (device fault, a persistent kernel's timeout) can leave arrival counters or accumulators non-zero,
    which would silently corrupt every later reducing / split-K call on that stream: callers that detect such an
    error call this AFTER synchronising the device."""
    for t in list(_STREAM_SCRATCH.values()) + list(_BN_SCRATCH.values()):
        t.zero_()


def conv_out_len(L, ks, stride, pad):
    return (L + 2 * pad - ks) // stride + 1


class HipKernels:
    """The product implementation: every method is one or a few HIP kernel launches."""

    name = "hip"

    # ---------------------------------------------------------------- conv1d
    def __init__(self):
        # packed weight images (include/m2d.h: m2d_conv1d_pack_weights) of live weight tensors; only
        # kept inside a weight_cache() scope, i.e. while the caller guarantees the weights are constant
        self._packed = {}
        self._cache_depth = 0
        self.pack_launches = 0

    @contextlib.contextmanager
    def weight_cache(self, keep=False):
        """Scope in which conv weights are promised constant except where invalidate_packed() is
        called (right after an optimizer step): the packed images the forward / backward-data
        GEMMs read are built once per weight tensor instead of once per call. Outside a scope every
        call packs afresh - a tensor's version counter is no proof of constancy (fused optimizers
        update parameters without moving it).
        keep: leave the images in place when the scope closes, for the next scope of the same owner (the
        training engine: the generator's weights do not change during the critic iterations of a cycle).
        The owner then vouches that between its scopes parameters only change through operations that move
        the version counter (copy_, load_state_dict, in-place math) or are announced with invalidate_packed()."""
        self._cache_depth += 1
        try:
            yield self
        finally:
            self._cache_depth -= 1
            if self._cache_depth == 0 and not keep:
                self._packed.clear()

    def invalidate_packed(self, tensors=None):
        """Drop the packed images of `tensors` (the parameters an optimizer step just changed), or all of them."""
        if tensors is None:
            self._packed.clear()
        else:
            kept = getattr(self, "_adam_kept", ())   # (w_fwd, w_bwd) images adam_multi has just rewritten stay
            for w in tensors:
                if id(w) not in kept:
                    self._packed.pop(id(w), None)
                self._packed.pop((id(w), "T"), None)
                for key in [k for k in self._packed if isinstance(k, tuple) and k[0] == id(w)]:
                    self._packed.pop(key, None)
            self._adam_kept = set()

    def packed_weights(self, w):
        """(w_fwd (Cin, ks, Cout), w_bwd (Cout, ks, Cin)): the K-major images of a conv weight the forward / backward-data
        GEMMs read (written on the current stream)."""
        stream = _stream(w.device)
        key = id(w)
        if self._cache_depth > 0:
            ent = self._packed.get(key)
            if ent is not None and ent[0]() is w and ent[1] == w._version and (ent[2] is None or ent[2] == stream):
                return ent[3], ent[4]
        Cout, Cin, ks = w.shape
        wf = torch.empty((Cin, ks, Cout), dtype=torch.float32, device=w.device)
        wb = torch.empty((Cout, ks, Cin), dtype=torch.float32, device=w.device)
        with _on(w.device):
            rc = _lib.lib().m2d_conv1d_pack_weights(_ptr(w), _ptr(wf), _ptr(wb), Cout, Cin, ks, stream)
        _lib.check(rc, "m2d_conv1d_pack_weights")
        self.pack_launches += 1
        if self._cache_depth > 0:
            packed = self._packed

            def _drop(_ref, key=key):
                packed.pop(key, None)

            packed[key] = (weakref.ref(w, _drop), w._version, stream, wf, wb)
        return wf, wb

    _SEPARATE_BIAS = os.environ.get("M2D_SEPARATE_BIAS", "1") != "0"
    _K4 = os.environ.get("M2D_K4", "1") != "0"  # A/B lever: 0 = stride-4 forwards through the generic engine
    _K4_OK = {}

    def _k4(self, Cin, L, Cout, ks, stride, pad):
        if not self._K4 or stride != 4:
            return False
        key = (Cin, L, Cout, ks, stride, pad)
        ok = self._K4_OK.get(key)
        if ok is None:
            ok = self._K4_OK[key] = bool(_lib.lib().m2d_conv1d_k4_applicable(Cin, L, Cout, ks, stride, pad))
        return ok

    def packed_k4(self, w, pad):
        """Wk4[(ci, tap group)][Cout][4]: the image the tap-vectorised stride-4 forward reads (cached like the others)."""
        stream = _stream(w.device)
        key = (id(w), "k4", pad)
        if self._cache_depth > 0:
            ent = self._packed.get(key)
            if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == stream:
                return ent[3]
        Cout, Cin, ks = w.shape
        h = _lib.lib()
        out = torch.empty((h.m2d_conv1d_k4_packed_elems(Cout, Cin, ks, pad),), dtype=torch.float32, device=w.device)
        with _on(w.device):
            rc = h.m2d_conv1d_pack_weights_k4(_ptr(w), _ptr(out), Cout, Cin, ks, pad, stream)
        _lib.check(rc, "m2d_conv1d_pack_weights_k4")
        self.pack_launches += 1
        if self._cache_depth > 0:
            packed = self._packed

            def _drop(_ref, key=key):
                packed.pop(key, None)

            packed[key] = (weakref.ref(w, _drop), w._version, stream, out)
        return out

    @staticmethod
    def _thin(Cin, ks, stride):
        return Cin == 1 and ks == 25 and stride == 4

    def conv1d_fwd(self, x, w, bias, stride, pad, act=0, slope=0.0, residual=None, out_mask=None,
                   out_mask_slope=0.0, with_stats=False, out=None, sum_out=None):
        """y = out_mask * act(conv(x) + bias) + residual  (mask BEFORE the residual)
        -> y, or (y, sums) with `with_stats`: sums (2*Cout,) float64 = per-channel sum / sum of squares of y
        from the conv's own epilogue (what a following BatchNorm needs: bn_fwd_sums).
        out: write y there (may alias out_mask: in-place masking). sum_out (needs residual): second output
        sum_out = y + residual while y itself stays without the residual; -> (y, sum_out)."""
        dev = _chk(x, w, bias, residual, out_mask, out, sum_out)
        B, Cin, L = x.shape
        Cout, Cin2, ks = w.shape
        assert Cin == Cin2, "conv1d: channel mismatch"
        Lout = conv_out_len(L, ks, stride, pad)
        y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=dev) if out is None else out
        assert tuple(y.shape) == (B, Cout, Lout)
        h = _lib.lib()
        full_length = Lout == 1 and pad == 0 and L == ks
        if sum_out is not None:
            assert not with_stats and residual is not None and tuple(sum_out.shape) == tuple(y.shape)
        if self._k4(Cin, L, Cout, ks, stride, pad):
            # tap-vectorised stride-4 forward (audio critic l2..l5 and the penalty's tangent through them)
            wk4 = self.packed_k4(w, pad)
            ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 0, B, Cin, L, Cout, ks, stride, pad), dev)
            sums = torch.empty((2 * Cout,), dtype=torch.float64, device=dev) if with_stats else None
            with _on(dev):
                rc = h.m2d_conv1d_fwd_k4(_ptr(x), _ptr(wk4), _ptr(bias), _ptr(y), _ptr(sum_out), B, Cin, L, Cout, ks,
                                         stride, pad, act, slope, _ptr(residual), _ptr(out_mask), out_mask_slope,
                                         _ptr(sums), _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream(dev))
            _lib.check(rc, "m2d_conv1d_fwd_k4")
            if sum_out is not None:
                return y, sum_out
            return (y, sums) if with_stats else y
        wp = self.packed_weights(w)[0] if (Cin >= 16 and not full_length) else None
        ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 0, B, Cin, L, Cout, ks, stride, pad), dev)
        nws = 0 if ws is None else ws.numel() * 4
        if sum_out is not None:
            with _on(dev):
                rc = h.m2d_conv1d_fwd_sum(_ptr(x), _ptr(w), _ptr(wp), _ptr(bias), _ptr(y), _ptr(sum_out), B, Cin, L,
                                          Cout, ks, stride, pad, act, slope, _ptr(residual), _ptr(out_mask),
                                          out_mask_slope, _ptr(ws), nws, _stream(dev))
            _lib.check(rc, "m2d_conv1d_fwd_sum")
            return y, sum_out
        sums = torch.empty((2 * Cout,), dtype=torch.float64, device=dev) if with_stats else None
        with _on(dev):
            rc = h.m2d_conv1d_fwd(_ptr(x), _ptr(w), _ptr(wp), _ptr(bias), _ptr(y), B, Cin, L, Cout, ks, stride, pad,
                                  act, slope, _ptr(residual), _ptr(out_mask), out_mask_slope, _ptr(sums), _ptr(ws),
                                  nws, _stream(dev))
        _lib.check(rc, "m2d_conv1d_fwd")
        return (y, sums) if with_stats else y

    def conv1d_bwd_data(self, dy, w, L, stride, pad, dy_mask=None, dy_mask_slope=0.0, out_mask=None,
                        out_mask_slope=0.0, residual=None, out=None):
        """dx = out_mask * (conv^T(dy * dy_mask, w) + residual); out: write dx there.
        out_mask may hold only the first B / 2 .. B samples: the samples behind them then read the mask from its start
        again (include/m2d.h: m2d_conv1d_bwd_data_shared_mask; no dy_mask / residual in that form)."""
        dev = _chk(dy, w, dy_mask, out_mask, residual, out)
        B, Cout, Lout = dy.shape
        Cout2, Cin, ks = w.shape
        assert Cout == Cout2 and Lout == conv_out_len(L, ks, stride, pad)
        dx = torch.empty((B, Cin, L), dtype=torch.float32, device=dev) if out is None else out
        assert tuple(dx.shape) == (B, Cin, L)
        if out_mask is not None and out_mask.shape[0] != B:
            mb = out_mask.shape[0]
            assert tuple(out_mask.shape[1:]) == (Cin, L) and 2 * mb >= B and dy_mask is None and residual is None
            wp = self.packed_weights(w)[1]
            ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 1, B, Cin, L, Cout, ks, stride, pad), dev)
            with _on(dev):
                rc = _lib.lib().m2d_conv1d_bwd_data_shared_mask(_ptr(dy), _ptr(w), _ptr(wp), _ptr(dx), B, Cin, L, Cout, ks,
                                                                stride, pad, _ptr(out_mask), out_mask_slope, mb, _ptr(ws),
                                                                0 if ws is None else ws.numel() * 4, _stream(dev))
            _lib.check(rc, "m2d_conv1d_bwd_data_shared_mask")
            return dx
        h = _lib.lib()
        full_length = Lout == 1 and pad == 0 and L == ks
        if (out_mask is not None or residual is not None) and self._thin(Cin, ks, stride):
            raise _lib.M2dError("conv1d_bwd_data: out_mask / residual are not supported on the thin (Cin = 1, k25 s4) path")
        wp = None if (full_length or self._thin(Cin, ks, stride)) else self.packed_weights(w)[1]
        ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 1, B, Cin, L, Cout, ks, stride, pad), dev)
        nws = 0 if ws is None else ws.numel() * 4
        with _on(dev):
            if residual is not None:
                assert tuple(residual.shape) == tuple(dx.shape)
                rc = h.m2d_conv1d_bwd_data_res(_ptr(dy), _ptr(w), _ptr(wp), _ptr(dx), B, Cin, L, Cout, ks, stride, pad,
                                               _ptr(dy_mask), dy_mask_slope, _ptr(residual), _ptr(out_mask),
                                               out_mask_slope, _ptr(ws), nws, _stream(dev))
            else:
                rc = h.m2d_conv1d_bwd_data(_ptr(dy), _ptr(w), _ptr(wp), _ptr(dx), B, Cin, L, Cout, ks, stride, pad,
                                           _ptr(dy_mask), dy_mask_slope, _ptr(out_mask), out_mask_slope, _ptr(ws),
                                           nws, _stream(dev))
        _lib.check(rc, "m2d_conv1d_bwd_data")
        return dx

    def conv1d_bwd_weight(self, x, dy, ks, stride, pad, dy_mask=None, dy_mask_slope=0.0, with_bias=False,
                          bias_from_sample=0):
        """-> dw, or (dw, dbias) with `with_bias`: dbias = sum over (batch, length) of the masked dy, over the
        samples [bias_from_sample, B) only (rows in front of them pair second-order operands: no bias term)."""
        dev = _chk(x, dy, dy_mask)
        B, Cin, L = x.shape
        B2, Cout, Lout = dy.shape
        assert B == B2 and Lout == conv_out_len(L, ks, stride, pad)
        dw = torch.empty((Cout, Cin, ks), dtype=torch.float32, device=dev)
        # The library takes the bias gradient from the same launch as one all-ones column of the x operand. Where
        # Cin * ks is a multiple of the 128-column tile that column costs a whole extra N tile (the encoder's k4 layers:
        # 129 columns = 2 tiles for 128, 257 = 3 for 256 ...), and for short outputs (K = (l, n) order: both operands
        # row-fast) it also keeps the launch off the LDS-direct kernel (measured in the step: the 128 -> 256 encoder
        # layer's weight gradient 501 us with the column, 190-220 us without). There the bias gradient is one
        # channel-sum pass over dy instead (M2D_SEPARATE_BIAS=0: always the column) - for short outputs and for up to four
        # column tiles; measured not to pay on the critic's layers (profiles/r05_separate_bias_shapes_diff.txt: the pose
        # critic's 896 + 1 columns 95.5 -> 89.2 us + a 22 us pass, the audio critic's 3 200 + 1 columns 567 -> 604 us).
        # (round 6: also for up to eight column tiles when the launch is long - the U-Net's 256 -> 128 k3 blocks at
        # 4 800 x 200 positions: 768 + 1 columns ran 7 tiles for 6.008, 2.46 ms at 77 TFLOP/s; the sum pass costs 0.12 ms)
        sep_bias = (with_bias and self._SEPARATE_BIAS and (Cin * ks) % 128 == 0 and
                    (Lout < 16 or Cin * ks <= 512 or (Cin * ks <= 1024 and B * Lout >= 200000))
                    and not self._thin(Cin, ks, stride) and _bn_scratch(dev, Cout) is not None)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev) if (with_bias and not sep_bias) else None
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 2, B, Cin, L, Cout, ks, stride, pad), dev)
        nws = 0 if ws is None else ws.numel() * 4
        with _on(dev):
            if bias_from_sample:
                rc = h.m2d_conv1d_bwd_weight_from(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), B, Cin, L, Cout, ks, stride,
                                                  pad, _ptr(dy_mask), dy_mask_slope, int(bias_from_sample), _ptr(ws),
                                                  nws, _stream(dev))
            else:
                rc = h.m2d_conv1d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), B, Cin, L, Cout, ks, stride, pad,
                                             _ptr(dy_mask), dy_mask_slope, _ptr(ws), nws, _stream(dev))
        _lib.check(rc, "m2d_conv1d_bwd_weight")
        if sep_bias:
            f = int(bias_from_sample)
            db = self.channel_sums(dy[f:] if f else dy, None if dy_mask is None else (dy_mask[f:] if f else dy_mask), dy_mask_slope)
        return (dw, db) if with_bias else dw

    # ---------------------------------------------------------------- conv over the windows of a padded track
    def conv1d_fwd_windows(self, track, T, hop, window, w, bias, stride, pad, act=0, slope=0.0, with_stats=False):
        """track (B, S); logical input (B*T, 1, window), window t of track b = track[b, t*hop : t*hop + window]
        (never materialised). -> y (B*T, Cout, Lout)."""
        dev = _chk(w, bias)
        B, S = _chk_track(track, dev, T, hop, window)
        Cout, Cin, ks = w.shape
        assert Cin == 1, "window views are single-channel"
        Lout = conv_out_len(window, ks, stride, pad)
        y = torch.empty((B * T, Cout, Lout), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 0, B * T, 1, window, Cout, ks, stride, pad), dev)
        sums = torch.empty((2 * Cout,), dtype=torch.float64, device=dev) if with_stats else None
        with _on(dev):
            rc = h.m2d_conv1d_fwd_windows(_ptr(track), B, S, T, hop, window, _ptr(w), _ptr(bias), _ptr(y), Cout, ks,
                                          stride, pad, act, slope, _ptr(sums), _ptr(ws),
                                          0 if ws is None else ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_conv1d_fwd_windows")
        return (y, sums) if with_stats else y

    def conv1d_bwd_weight_windows(self, track, T, hop, window, dy, ks, stride, pad, dy_mask=None, dy_mask_slope=0.0,
                                  with_bias=False):
        dev = _chk(dy, dy_mask)
        B, S = _chk_track(track, dev, T, hop, window)
        N, Cout, Lout = dy.shape
        assert N == B * T and Lout == conv_out_len(window, ks, stride, pad)
        dw = torch.empty((Cout, 1, ks), dtype=torch.float32, device=dev)
        db = torch.empty((Cout,), dtype=torch.float32, device=dev) if with_bias else None
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_conv1d_workspace_bytes', 2, B * T, 1, window, Cout, ks, stride, pad), dev)
        with _on(dev):
            rc = h.m2d_conv1d_bwd_weight_windows(_ptr(track), B, S, T, hop, window, _ptr(dy), _ptr(dw), _ptr(db), Cout,
                                                 ks, stride, pad, _ptr(dy_mask), dy_mask_slope, _ptr(ws),
                                                 0 if ws is None else ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_conv1d_bwd_weight_windows")
        return (dw, db) if with_bias else dw

    # ---------------------------------------------------------------- gemm
    def gemm(self, mode, a, b, bias=None, act=0, slope=0.0, a_mask=None, a_mask_slope=0.0, out_mask=None,
             out_mask_slope=0.0, out=None):
        """mode 0: a(M,K) b(N,K)^T (+bias[N]); 1: a(M,K) b(K,N); 2: a(K,M)^T b(K,N). out: write there (may alias
        out_mask: in-place masking)."""
        dev = _chk(a, b, bias, a_mask, out_mask, out)
        if mode == 0:
            M, K = a.shape
            N, K2 = b.shape
        elif mode == 1:
            M, K = a.shape
            K2, N = b.shape
        else:
            K, M = a.shape
            K2, N = b.shape
        assert K == K2, "gemm: inner dimension mismatch"
        c = torch.empty((M, N), dtype=torch.float32, device=dev) if out is None else out
        assert tuple(c.shape) == (M, N)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_gemm_workspace_bytes', mode, M, N, K), dev)
        with _on(dev):
            rc = h.m2d_gemm(mode, _ptr(a), _ptr(b), _ptr(bias), _ptr(c), M, N, K, act, slope, _ptr(a_mask),
                            a_mask_slope, _ptr(out_mask), out_mask_slope, _ptr(ws),
                            0 if ws is None else ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_gemm")
        return c

    @staticmethod
    def _chk_rows(*tensors):
        """2-D fp32 device tensors whose rows are dense (stride(1) == 1) but may be a column block of a wider
        buffer (any stride(0) >= the row length) -> device."""
        dev = None
        for t in tensors:
            if t is None:
                continue
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
                raise _lib.M2dError("gemm_ld: 2-D fp32 HIP tensors only")
            if t.size(1) > 1 and t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
                raise _lib.M2dError("gemm_ld: rows must be dense")
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise _lib.M2dError("tensors on different devices")
        return dev

    @staticmethod
    def _ld(t):
        return t.stride(0) if t.size(0) > 1 else t.size(1)

    def gemm_ld(self, mode, a, b, bias=None, act=0, slope=0.0, a_mask=None, a_mask_slope=0.0, out_mask=None,
                out_mask_slope=0.0, out=None):
        """gemm() on row-strided 2-D views (column blocks of wider buffers): a_mask must share a's layout,
        out_mask the output's (it may BE the output: in-place masking). out: a view to write into."""
        dev = self._chk_rows(a, b, a_mask, out_mask, out)
        _chk(bias)
        if mode == 0:
            M, K = a.shape
            N, K2 = b.shape
        elif mode == 1:
            M, K = a.shape
            K2, N = b.shape
        else:
            K, M = a.shape
            K2, N = b.shape
        assert K == K2, "gemm: inner dimension mismatch"
        c = torch.empty((M, N), dtype=torch.float32, device=dev) if out is None else out
        assert tuple(c.shape) == (M, N)
        if a_mask is not None:
            assert a_mask.shape == a.shape and self._ld(a_mask) == self._ld(a)
        if out_mask is not None:
            assert out_mask.shape == c.shape and self._ld(out_mask) == self._ld(c)
        ws = _ws(_ws_bytes('m2d_gemm_workspace_bytes', mode, M, N, K), dev)
        with _on(dev):
            rc = _lib.lib().m2d_gemm_ld(mode, _ptr(a), self._ld(a), _ptr(b), self._ld(b), _ptr(bias), _ptr(c),
                                        self._ld(c), M, N, K, act, slope, _ptr(a_mask), a_mask_slope, _ptr(out_mask),
                                        out_mask_slope, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_gemm_ld")
        return c

    # ---------------------------------------------------------------- tanh heads
    def tanh_fwd(self, x):
        dev = _chk(x)
        y = torch.empty_like(x)
        with _on(dev):
            rc = _lib.lib().m2d_tanh_fwd(_ptr(x), _ptr(y), x.numel(), _stream(dev))
        _lib.check(rc, "m2d_tanh_fwd")
        return y

    def tanh_bwd(self, gy, y):
        """gy * (1 - y^2)"""
        dev = _chk(gy, y)
        gx = torch.empty_like(gy)
        with _on(dev):
            rc = _lib.lib().m2d_tanh_bwd(_ptr(gy), _ptr(y), _ptr(gx), gy.numel(), _stream(dev))
        _lib.check(rc, "m2d_tanh_bwd")
        return gx

    def tanh_bwd_bwd(self, g, gy, y):
        """-2 * y * g * gy: the gradient of tanh_bwd(gy, y) w.r.t. y for cotangent g"""
        dev = _chk(g, gy, y)
        out = torch.empty_like(y)
        with _on(dev):
            rc = _lib.lib().m2d_tanh_bwd_bwd(_ptr(g), _ptr(gy), _ptr(y), _ptr(out), y.numel(), _stream(dev))
        _lib.check(rc, "m2d_tanh_bwd_bwd")
        return out

    def adam_multi(self, params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step, skip=None, repack=True):
        """One torch.optim.Adam step (defaults: no weight decay, no amsgrad) over the listed tensors in one launch per 48
        (include/m2d.h: m2d_adam_multi). repack: conv weights (3-D, >= 16 input channels) that have live packed images in
        the weight cache get those images rewritten in the same pass - the cache entries stay valid across the step
        (the engines call invalidate_packed() for everything else). skip: optional device float (non-zero = no-op)."""
        if not params:
            return
        dev = _chk(*params, *grads, *exp_avgs, *exp_avg_sqs, skip if torch.is_tensor(skip) else None)
        items = (_lib.AdamItem * len(params))()
        stream = _stream(dev)
        keep = []
        for it, p, g, m, v in zip(items, params, grads, exp_avgs, exp_avg_sqs):
            assert p.numel() == g.numel() == m.numel() == v.numel()
            it.param, it.grad, it.exp_avg, it.exp_avg_sq, it.numel = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            ent = self._packed.get(id(p)) if (repack and p.dim() == 3) else None
            if ent is not None and ent[0]() is p:
                # written on THIS stream; consumers on other streams fork behind it (side.wait_stream(main) in the
                # modules), so the refreshed entry is marked valid for every stream (None)
                wf, wb = ent[3], ent[4]
                it.pack_fwd, it.pack_bwd = wf.data_ptr(), wb.data_ptr()
                it.cout, it.cin, it.ks = p.shape
                keep.append((id(p), (ent[0], p._version, None, wf, wb)))
        bc1 = 1.0 - beta1 ** step
        bc2s = (1.0 - beta2 ** step) ** 0.5
        with _on(dev):
            rc = _lib.lib().m2d_adam_multi(ctypes.addressof(items), len(params), lr, beta1, beta2, eps, bc1, bc2s,
                                           (skip if isinstance(skip, int) else _ptr(skip)) or 0, stream)
        _lib.check(rc, "m2d_adam_multi")
        for key, ent in keep:
            self._packed[key] = ent
        # (accumulated over the launches of one optimizer step - several step groups / 48-tensor batches - and consumed
        # by the invalidate_packed() that follows the step)
        self._adam_kept = set(getattr(self, "_adam_kept", ())) | {k for k, _ in keep}

    def transposed(self, w):
        """w.t().contiguous() of a 2-D weight (the GRU kernels read W^T), kept like the packed conv images: built
        once per weight tensor inside a weight_cache() scope, dropped by invalidate_packed()."""
        stream = _stream(w.device)
        key = (id(w), "T")
        if self._cache_depth > 0:
            ent = self._packed.get(key)
            if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == stream:
                return ent[3]
        wt = w.t().contiguous()
        if self._cache_depth > 0:
            packed = self._packed

            def _drop(_ref, key=key):
                packed.pop(key, None)

            packed[key] = (weakref.ref(w, _drop), w._version, stream, wt)
        return wt

    # ---------------------------------------------------------------- critic iteration: pack + loss
    def pose_pack3(self, real, fake_rows, alpha, out=None):
        """real (B, T, C), fake_rows (B*T, C), alpha (B,) -> (3B, C, T) = [interpolated | real | fake] channels-first."""
        dev = _chk(real, fake_rows, alpha, out)
        B, T, C = real.shape
        assert fake_rows.numel() == real.numel() and alpha.numel() == B
        o = torch.empty((3 * B, C, T), dtype=torch.float32, device=dev) if out is None else out
        with _on(dev):
            rc = _lib.lib().m2d_pose_pack3(_ptr(real), _ptr(fake_rows), _ptr(alpha), _ptr(o), B, T, C, _stream(dev))
        _lib.check(rc, "m2d_pose_pack3")
        return o

    def wgan_critic_loss(self, scores, B, pen0, pen1, gamma):
        """scores (3B,) = [interpolated | real | fake] -> (3,) = (loss_critic, gp, w_dist)."""
        dev = _chk(scores, pen0, pen1)
        assert scores.numel() == 3 * B
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_wgan_critic_loss(_ptr(scores), B, _ptr(pen0), _ptr(pen1), float(gamma), _ptr(out),
                                                 _stream(dev))
        _lib.check(rc, "m2d_wgan_critic_loss")
        return out

    # ---------------------------------------------------------------- batch norm
    def channel_sums(self, x, mask=None, slope=0.0):
        """x: (B, C, L) or (B, C) -> (C,) sums over batch and length (optionally masked)."""
        dev = _chk(x, mask)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        out = torch.empty((C,), dtype=torch.float32, device=dev)
        h = _lib.lib()
        scratch = _bn_scratch(dev, C)
        # no zero-kept scratch (C wider than the fixed one, or a capture that was not given one): the stateless
        # three-launch form needs its workspace (passing none made the call fail with "workspace too small")
        ws = None if scratch is not None else _ws(_ws_bytes('m2d_bn_workspace_bytes', C), dev)
        with _on(dev):
            rc = h.m2d_channel_sums(_ptr(x), _ptr(mask), slope, _ptr(out), B, C, L, _ptr(ws),
                                    0 if ws is None else ws.numel() * 4, _ptr(scratch), _stream(dev))
        _lib.check(rc, "m2d_channel_sums")
        return out

    @staticmethod
    def _out_block(out, like):
        """`out`: where a (B, C, L) result goes when it is a channel block of a wider buffer - a view shaped like the
        result whose samples are out.stride(0) elements apart, dense inside a sample. -> (tensor, batch stride)"""
        if out is None:
            return torch.empty_like(like), 0
        assert out.shape == like.shape and out.dtype == torch.float32 and out.device == like.device
        assert out.dim() == 3 and out.stride(2) == 1 and out.stride(1) == out.shape[2] and out.stride(0) >= out.shape[1] * out.shape[2]
        return out, out.stride(0)

    def bn_fwd(self, x, gamma, beta, running_mean, running_var, training, eps, momentum, act=0, slope=0.0,
               residual=None, out=None):
        dev = _chk(x, gamma, beta, running_mean, running_var, residual)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        y, ypitch = self._out_block(out, x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_bn_workspace_bytes', C), dev)
        with _on(dev):
            rc = h.m2d_bn_fwd_to(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(y),
                                 _ptr(save_mean), _ptr(save_invstd), B, C, L, eps, momentum, 1 if training else 0,
                                 act, slope, _ptr(residual), _ptr(ws), ws.numel() * 4,
                                 _ptr(_bn_scratch(dev, C)) if training else 0, ypitch, _stream(dev))
        _lib.check(rc, "m2d_bn_fwd")
        return y, save_mean, save_invstd

    @staticmethod
    def _sums_ok(sums, C, dev):
        if sums.dtype != torch.float64 or sums.numel() != 2 * C or not sums.is_contiguous() or sums.device != dev:
            raise _lib.M2dError("BatchNorm sums must be a contiguous float64 tensor of 2*C elements on the device")

    def bn_stats(self, x):
        """(2C,) float64: per-channel sum and sum of squares of x (B, C[, L]) over batch and length."""
        dev = _chk(x)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        sums = torch.empty((2 * C,), dtype=torch.float64, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_stats(_ptr(x), _ptr(sums), B, C, L, _ptr(_bn_scratch(dev, C)), _stream(dev))
        _lib.check(rc, "m2d_bn_stats")
        return sums

    def bn_fwd_sums(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                    residual=None, out=None):
        """Training forward from batch sums over `count` elements per channel (bn_stats, a conv's epilogue, or
        their all-reduce across data-parallel ranks). -> y, save_mean, save_invstd."""
        dev = _chk(x, gamma, beta, running_mean, running_var, residual)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        self._sums_ok(sums, C, dev)
        y, ypitch = self._out_block(out, x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_fwd_sums_to(_ptr(x), _ptr(sums), float(count), _ptr(gamma), _ptr(beta),
                                               _ptr(running_mean), _ptr(running_var), _ptr(y), _ptr(save_mean),
                                               _ptr(save_invstd), B, C, L, eps, momentum, act, slope, _ptr(residual),
                                               ypitch, _stream(dev))
        _lib.check(rc, "m2d_bn_fwd_sums")
        return y, save_mean, save_invstd

    def bn_update_running(self, sums, count, running_mean, running_var, eps, momentum):
        """running buffers advanced once more with batch sums they were already advanced with (include/m2d.h)"""
        dev = _chk(running_mean, running_var)
        C = running_mean.numel()
        self._sums_ok(sums, C, dev)
        tmp = torch.empty((2 * C,), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_update_running(_ptr(sums), float(count), _ptr(running_mean), _ptr(running_var),
                                                  _ptr(tmp), C, eps, momentum, _stream(dev))
        _lib.check(rc, "m2d_bn_update_running")

    def bn_fwd_sums_pool(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                         out=None):
        """bn_fwd_sums with MaxPool1d(2, 2) of the result made in the same pass. -> (y, pooled, save_mean, save_invstd);
        out: where y goes (a channel block of a wider buffer, as bn_fwd_sums)."""
        dev = _chk(x, gamma, beta, running_mean, running_var)
        B, C, L = x.shape
        self._sums_ok(sums, C, dev)
        y, ypitch = self._out_block(out, x)
        pooled = torch.empty((B, C, L // 2), dtype=torch.float32, device=dev)
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_fwd_sums_pool_to(_ptr(x), _ptr(sums), float(count), _ptr(gamma), _ptr(beta),
                                                    _ptr(running_mean), _ptr(running_var), _ptr(y), _ptr(pooled),
                                                    _ptr(save_mean), _ptr(save_invstd), B, C, L, eps, momentum, act, slope,
                                                    ypitch, _stream(dev))
        _lib.check(rc, "m2d_bn_fwd_sums_pool_to")
        return y, pooled, save_mean, save_invstd

    def bn_fwd_sums_upsample2(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                              out=None):
        """upsample2_linear(bn_fwd_sums(x)) in one pass (the normalised tensor itself is not written).
        -> (up (B, C, 2L), save_mean, save_invstd); out: a (B, C, 2L) channel block of a wider buffer."""
        dev = _chk(x, gamma, beta, running_mean, running_var)
        B, C, L = x.shape
        self._sums_ok(sums, C, dev)
        if out is None:
            up, pitch = torch.empty((B, C, 2 * L), dtype=torch.float32, device=dev), 0
        else:
            assert out.shape == (B, C, 2 * L) and out.dtype == torch.float32 and out.device == x.device
            assert out.stride(2) == 1 and out.stride(1) == 2 * L and out.stride(0) >= 2 * C * L
            up, pitch = out, out.stride(0)
        save_mean = torch.empty((C,), dtype=torch.float32, device=dev)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_fwd_sums_upsample2_to(_ptr(x), _ptr(sums), float(count), _ptr(gamma), _ptr(beta),
                                                         _ptr(running_mean), _ptr(running_var), _ptr(up), _ptr(save_mean),
                                                         _ptr(save_invstd), B, C, L, eps, momentum, act, slope, pitch,
                                                         _stream(dev))
        _lib.check(rc, "m2d_bn_fwd_sums_upsample2_to")
        return up, save_mean, save_invstd

    def bn_bwd_stats(self, dy, x, gamma, beta, save_mean, save_invstd, act=0, slope=0.0):
        """(2C,) float64: sum dz and sum dz * xhat, dz = dy * act'(bn(x))."""
        dev = _chk(dy, x, gamma, beta, save_mean, save_invstd)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        sums = torch.empty((2 * C,), dtype=torch.float64, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_bwd_stats(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                             _ptr(save_invstd), _ptr(sums), B, C, L, act, slope,
                                             _ptr(_bn_scratch(dev, C)), _stream(dev))
        _lib.check(rc, "m2d_bn_bwd_stats")
        return sums

    def bn_bwd_sums(self, dy, x, gamma, beta, save_mean, save_invstd, sums_local, sums_global, count, act=0, slope=0.0):
        dev = _chk(dy, x, gamma, beta, save_mean, save_invstd)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        self._sums_ok(sums_local, C, dev), self._sums_ok(sums_global, C, dev)
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        ws = _ws(_ws_bytes('m2d_bn_workspace_bytes', C), dev)
        with _on(dev):
            rc = _lib.lib().m2d_bn_bwd_sums(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                            _ptr(save_invstd), _ptr(sums_local), _ptr(sums_global), float(count),
                                            _ptr(dx), _ptr(dgamma), _ptr(dbeta), B, C, L, act, slope, _ptr(ws),
                                            ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_bn_bwd_sums")
        return dx, dgamma, dbeta

    def bn_bwd(self, dy, x, gamma, beta, save_mean, save_invstd, act=0, slope=0.0):
        dev = _chk(dy, x, gamma, beta, save_mean, save_invstd)
        B, C = x.shape[0], x.shape[1]
        L = x.shape[2] if x.dim() == 3 else 1
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_bn_workspace_bytes', C), dev)
        with _on(dev):
            rc = h.m2d_bn_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(save_mean), _ptr(save_invstd),
                              _ptr(dx), _ptr(dgamma), _ptr(dbeta), B, C, L, act, slope, _ptr(ws), ws.numel() * 4,
                              _ptr(_bn_scratch(dev, C)), _stream(dev))
        _lib.check(rc, "m2d_bn_bwd")
        return dx, dgamma, dbeta

    # ---------------------------------------------------------------- GRU
    def gru_layer_fwd(self, gi, w_hh_t, b_hh, lengths=None, save=True):
        """gi: (B, T, 3H) with b_ih added; w_hh_t: (H, 3H). Returns out (B,T,H), saved gates or None."""
        dev = _chk(gi, w_hh_t, b_hh)
        B, T, H3 = gi.shape
        H = H3 // 3
        out = torch.empty((B, T, H), dtype=torch.float32, device=dev)
        saved = None
        if save:
            saved = torch.empty((4, B, T, H), dtype=torch.float32, device=dev)
        h = _lib.lib()
        with _on(dev):
            rc = h.m2d_gru_layer_fwd(_ptr(gi), _ptr(w_hh_t), _ptr(b_hh), _ptr(lengths), _ptr(out),
                                     _ptr(saved[0]) if save else 0, _ptr(saved[1]) if save else 0,
                                     _ptr(saved[2]) if save else 0, _ptr(saved[3]) if save else 0, B, T, H,
                                     _stream(dev))
        _lib.check(rc, "m2d_gru_layer_fwd")
        return out, saved

    def gru_layer_bwd(self, dout, out, saved, w_hh, lengths=None):
        dev = _chk(dout, out, saved, w_hh)
        B, T, H = out.shape
        dgi = torch.empty((B, T, 3 * H), dtype=torch.float32, device=dev)
        dgh = torch.empty((B, T, 3 * H), dtype=torch.float32, device=dev)
        dh_buf = torch.empty((2, B, H), dtype=torch.float32, device=dev)
        h = _lib.lib()
        with _on(dev):
            rc = h.m2d_gru_layer_bwd(_ptr(dout), _ptr(out), _ptr(saved[0]), _ptr(saved[1]), _ptr(saved[2]),
                                     _ptr(saved[3]), _ptr(w_hh), _ptr(lengths), _ptr(dgi), _ptr(dgh),
                                     _ptr(dh_buf), B, T, H, _stream(dev))
        _lib.check(rc, "m2d_gru_layer_bwd")
        return dgi, dgh

    @staticmethod
    def _ptr_array(tensors):
        arr = (ctypes.c_void_p * len(tensors))(*[_ptr(t) for t in tensors])
        return arr, ctypes.cast(arr, ctypes.c_void_p)

    def gru_stack_fwd(self, gi0, w_ih_t, b_ih, w_hh_t, b_hh, lengths=None, save=True, persistent=True):
        """L-layer GRU on the (layer, t) diagonal. Lists have L entries (entry 0 of w_ih_t / b_ih may
        be None). Returns ([out_l (B,T,H)], [saved_l (4,B,T,H)] or None)."""
        L = len(w_hh_t)
        dev = _chk(gi0, *[t for t in list(w_ih_t) + list(b_ih) + list(w_hh_t) + list(b_hh) if t is not None])
        B, T, H3 = gi0.shape
        H = H3 // 3
        # (one allocation: the persistent launch pre-fills every layer's output with its hand-off sentinel in one memset)
        outs = list(torch.empty((L, B, T, H), dtype=torch.float32, device=dev).unbind(0))
        saved = [torch.empty((4, B, T, H), dtype=torch.float32, device=dev) for _ in range(L)] if save else None
        keep = [self._ptr_array(v) for v in (w_ih_t, b_ih, w_hh_t, b_hh, outs)]
        sv = self._ptr_array(saved) if save else (None, None)
        h = _lib.lib()
        # scratch for the persistent form (one launch for the whole recurrence); the library decides
        counters = (torch.empty((h.m2d_gru_stack_counters(B, L),), dtype=torch.int32, device=dev)
                    if (persistent and self.persistent_gru) else None)
        with _on(dev):
            rc = h.m2d_gru_stack_fwd(_ptr(gi0), keep[0][1], keep[1][1], keep[2][1], keep[3][1], keep[4][1],
                                     sv[1], _ptr(lengths), B, T, H, L, _ptr(counters), _stream(dev))
        _lib.check(rc, "m2d_gru_stack_fwd")
        return outs, saved

    @staticmethod
    def check_async_errors():
        """Raise if a persistent GRU launch timed out (another process holding the CUs it waits for).
        Call after a stream synchronisation; cheap (reads one pinned host word)."""
        if _lib.lib().m2d_gru_persist_error():
            raise _lib.M2dError("persistent GRU launch timed out: outputs of that call are invalid "
                                "(several processes on one GPU? set M2D_PERSISTENT_GRU=0)")

    # -- recovery from such a timeout inside the process (engine.WganGpEngine._check_async)
    persistent_gru = True        # False: every recurrence from now on runs as per-step launches
    async_faults = 0             # timeouts recovered from

    @staticmethod
    def fault_word():
        """Device-visible address of the word a persistent recurrence raises when it gives up (or None): the `skip`
        argument of adam_multi - an optimizer step queued behind the failed launch then voids itself on the device."""
        return _lib.lib().m2d_async_fault_word() or None

    @staticmethod
    def fault_fetch(dst):
        """dst[0] (device float) = 1.0 when that word is raised, else 0.0, on the current stream (dp.GradExchange)"""
        dev = _chk(dst)
        with _on(dev):
            _lib.check(_lib.lib().m2d_fault_fetch(_ptr(dst), _stream(dev)), "m2d_fault_fetch")

    @staticmethod
    def raise_async_fault():
        """test hook: raise the word as a timed-out recurrence would"""
        _lib.check(_lib.lib().m2d_gru_persist_raise(), "m2d_gru_persist_raise")

    @classmethod
    def recover_async_fault(cls):
        """-> True when a persistent recurrence had timed out. The word is still raised at this point, so after the
        device synchronisation made here EVERY optimizer step queued behind the failed launch has skipped itself (no
        garbage reached the parameters); then the word is cleared, the scratch re-zeroed, and recurrences run as
        per-step launches from now on (fresh launches - nothing is re-executed, the affected iterations are lost)."""
        if not _lib.lib().m2d_gru_persist_peek():
            return False
        torch.cuda.synchronize()
        _lib.lib().m2d_gru_persist_error()
        reset_scratch()
        cls.persistent_gru = False
        cls.async_faults += 1
        return True

    def gru_stack_bwd(self, dout, outs, saved, w_hh, w_ih, lengths=None, persistent=True):
        """BPTT of the stack; returns ([dgi_l (B,T,3H)], [dgh_l (B,T,3H)]). persistent: one launch for the whole
        recurrence when the library finds room for it (bit-identical to the step launches)."""
        L = len(outs)
        dev = _chk(dout, *outs, *saved, *w_hh, *[t for t in w_ih if t is not None])
        B, T, H = outs[0].shape
        dgi = [torch.empty((B, T, 3 * H), dtype=torch.float32, device=dev) for _ in range(L)]
        dgh = [torch.empty((B, T, 3 * H), dtype=torch.float32, device=dev) for _ in range(L)]
        dhb = [torch.empty((2, B, H), dtype=torch.float32, device=dev) for _ in range(L)]
        keep = [self._ptr_array(v) for v in (outs, saved, w_hh, w_ih, dgi, dgh, dhb)]
        h = _lib.lib()
        counters = (torch.empty((h.m2d_gru_stack_counters(B, L),), dtype=torch.int32, device=dev)
                    if (persistent and self.persistent_gru) else None)
        with _on(dev):
            rc = h.m2d_gru_stack_bwd(_ptr(dout), keep[0][1], keep[1][1], keep[2][1], keep[3][1], keep[4][1],
                                     keep[5][1], keep[6][1], _ptr(lengths), B, T, H, L, _ptr(counters), _stream(dev))
        _lib.check(rc, "m2d_gru_stack_bwd")
        return dgi, dgh

    # ---------------------------------------------------------------- gradient penalty
    def gp_interpolate(self, real, fake, alpha):
        """real, fake: (B, n); alpha: (B,) -> alpha*real + (1-alpha)*fake."""
        dev = _chk(real, fake, alpha)
        B, n = real.shape
        out = torch.empty_like(real)
        with _on(dev):
            rc = _lib.lib().m2d_gp_interpolate(_ptr(real), _ptr(fake), _ptr(alpha), _ptr(out), B, n, _stream(dev))
        _lib.check(rc, "m2d_gp_interpolate")
        return out

    def gp_penalty_fwd(self, g, lp):
        dev = _chk(g)
        B, n = g.shape
        norms = torch.empty((B,), dtype=torch.float32, device=dev)
        pen = torch.empty((), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_gp_penalty_workspace_bytes', B), dev)
        with _on(dev):
            rc = h.m2d_gp_penalty_fwd(_ptr(g), _ptr(norms), _ptr(pen), B, n, 1 if lp else 0, _ptr(ws),
                                      ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_gp_penalty_fwd")
        return pen, norms

    def gp_penalty_bwd(self, g, norms, gout, lp, out=None):
        dev = _chk(g, norms, gout, out)
        B, n = g.shape
        dg = torch.empty_like(g) if out is None else out
        assert dg.numel() == g.numel()
        with _on(dev):
            rc = _lib.lib().m2d_gp_penalty_bwd(_ptr(g), _ptr(norms), _ptr(gout), _ptr(dg), B, n, 1 if lp else 0,
                                               _stream(dev))
        _lib.check(rc, "m2d_gp_penalty_bwd")
        return dg

    # ---------------------------------------------------------------- losses
    def l1_mean_fwd(self, a, b):
        dev = _chk(a, b)
        out = torch.empty((), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_reduce_workspace_bytes', ), dev)
        with _on(dev):
            rc = h.m2d_l1_mean_fwd(_ptr(a), _ptr(b), _ptr(out), a.numel(), _ptr(ws), ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_l1_mean_fwd")
        return out

    def l1_mean_bwd(self, a, b, gout):
        dev = _chk(a, b, gout)
        da = torch.empty_like(a)
        with _on(dev):
            rc = _lib.lib().m2d_l1_mean_bwd(_ptr(a), _ptr(b), _ptr(gout), _ptr(da), a.numel(), _stream(dev))
        _lib.check(rc, "m2d_l1_mean_bwd")
        return da

    def tv_mean_fwd(self, x, B, C, T, sb, sc, st):
        """x: storage holding a (B, C, T) view with element strides (sb, sc, st)."""
        dev = _chk(x)
        out = torch.empty((), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_reduce_workspace_bytes', ), dev)
        with _on(dev):
            rc = h.m2d_tv_mean_fwd(_ptr(x), _ptr(out), B, C, T, sb, sc, st, _ptr(ws), ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_tv_mean_fwd")
        return out

    def tv_mean_bwd(self, x, gout, B, C, T, sb, sc, st):
        dev = _chk(x, gout)
        dx = torch.empty_like(x)
        with _on(dev):
            rc = _lib.lib().m2d_tv_mean_bwd(_ptr(x), _ptr(gout), _ptr(dx), B, C, T, sb, sc, st, _stream(dev))
        _lib.check(rc, "m2d_tv_mean_bwd")
        return dx

    # ---------------------------------------------------------------- evaluation metric, dataset scaling
    def jerk_mean_fwd(self, x, B, C, T, sb, sc, st):
        """jerkiness of the (B, C, T) view (element strides sb, sc, st) of storage `x` -> 0-dim."""
        dev = _chk(x)
        out = torch.empty((), dtype=torch.float32, device=dev)
        h = _lib.lib()
        ws = _ws(_ws_bytes('m2d_reduce_workspace_bytes', ), dev)
        with _on(dev):
            rc = h.m2d_jerk_mean_fwd(_ptr(x), _ptr(out), B, C, T, sb, sc, st, _ptr(ws), ws.numel() * 4, _stream(dev))
        _lib.check(rc, "m2d_jerk_mean_fwd")
        return out

    def affine_cols(self, x, scale, shift, out=None):
        """x (..., cols) * scale[cols] + shift[cols] (MinMaxScaler transform / inverse); out may be x."""
        dev = _chk(x, scale, shift, out)
        cols = x.shape[-1]
        assert scale.numel() == cols and shift.numel() == cols
        y = torch.empty_like(x) if out is None else out
        with _on(dev):
            rc = _lib.lib().m2d_affine_cols(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), x.numel() // max(cols, 1), cols,
                                            _stream(dev))
        _lib.check(rc, "m2d_affine_cols")
        return y

    # ---------------------------------------------------------------- U-Net resampling
    def maxpool2_fwd(self, x):
        """x: (B, C, L) dense, or a channel block of a wider buffer (samples x.stride(0) apart, dense inside)"""
        B, C, L = x.shape
        if not x.is_contiguous() and x.stride(2) == 1 and x.stride(1) == L and x.stride(0) > C * L and x.is_cuda \
                and x.dtype == torch.float32 and L % 2 == 0 and (C * L) % 8 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
            dev = x.device
            y = torch.empty((B, C, L // 2), dtype=torch.float32, device=dev)
            with _on(dev):
                rc = _lib.lib().m2d_maxpool2_fwd_from(_ptr(x), _ptr(y), B, C, L, x.stride(0), _stream(dev))
            _lib.check(rc, "m2d_maxpool2_fwd_from")
            return y
        x = x if x.is_contiguous() else x.contiguous()
        dev = _chk(x)
        y = torch.empty((B, C, L // 2), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_maxpool2_fwd(_ptr(x), _ptr(y), B * C, L, _stream(dev))
        _lib.check(rc, "m2d_maxpool2_fwd")
        return y

    def maxpool2_bwd(self, x, dy):
        dev = _chk(x, dy)
        B, C, L = x.shape
        dx = torch.empty_like(x)
        with _on(dev):
            rc = _lib.lib().m2d_maxpool2_bwd(_ptr(x), _ptr(dy), _ptr(dx), B * C, L, _stream(dev))
        _lib.check(rc, "m2d_maxpool2_bwd")
        return dx

    def upsample2_fwd(self, x, out=None):
        dev = _chk(x)
        B, C, L = x.shape
        if out is None:
            y, ypitch = torch.empty((B, C, 2 * L), dtype=torch.float32, device=dev), 0
        else:
            assert tuple(out.shape) == (B, C, 2 * L) and out.dtype == torch.float32 and out.device == dev
            assert out.stride(2) == 1 and out.stride(1) == 2 * L and out.stride(0) >= 2 * C * L
            y, ypitch = out, out.stride(0)
        with _on(dev):
            rc = _lib.lib().m2d_upsample2_fwd_to(_ptr(x), _ptr(y), B, C, L, ypitch, _stream(dev))
        _lib.check(rc, "m2d_upsample2_fwd")
        return y

    def upsample2_bwd(self, dy):
        dev = _chk(dy)
        B, C, Lo = dy.shape
        dx = torch.empty((B, C, Lo // 2), dtype=torch.float32, device=dev)
        with _on(dev):
            rc = _lib.lib().m2d_upsample2_bwd(_ptr(dy), _ptr(dx), B * C, Lo // 2, _stream(dev))
        _lib.check(rc, "m2d_upsample2_bwd")
        return dx

    # ---------------------------------------------------------------- profiler
    def prof_begin(self):
        _lib.check(_lib.lib().m2d_prof_begin(), "m2d_prof_begin")

    def prof_dump(self):
        """[(family, tag, d0, d1, d2, ms, flops, bytes)] per launch of the current profiling session."""
        cap = 1 << 22
        buf = ctypes.create_string_buffer(cap)
        n = _lib.lib().m2d_prof_dump(buf, cap)
        rows = []
        for line in buf.raw[:n].decode().splitlines():
            f = line.split(",")
            rows.append((int(f[0]), f[1], int(f[2]), int(f[3]), int(f[4]), float(f[5]), float(f[6]),
                         float(f[7]) if len(f) > 7 else 0.0))
        return rows

    def prof_end(self):
        buf = (ctypes.c_double * 20)()
        _lib.check(_lib.lib().m2d_prof_end(buf, 20), "m2d_prof_end")
        fams = ["gemm", "bn", "gru", "pointwise", "reduce"]
        return {f: {"ms": buf[4 * i], "launches": int(buf[4 * i + 1]), "flops": buf[4 * i + 2],
                    "bytes": buf[4 * i + 3]} for i, f in enumerate(fams)}


_impl = HipKernels()


def impl():
    return _impl


def set_impl(obj):
    """Test hook (tests/fake_backend.py installs a CPU stand-in to exercise host logic)."""
    global _impl
    prev = _impl
    _impl = obj
    return prev
