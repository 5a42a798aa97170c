"""Test infrastructure: which activation masks does the product's forward use, and are the ones that differ from the
oracle's rounding flips - and nothing else?

Two fp32 evaluations of a ReLU / LeakyReLU network (the product's kernels, the oracle's torch ops) do not share all
activation masks: a pre-activation within rounding of zero is positive in one and negative in the other. Each such flip
moves the gradients it touches by O(1) of their size, which is why gradient comparisons across the two used to carry a
slack term (tests/test_gpu_full_size_parity.py, rounds 3-5). This module makes the comparison exact instead:

  product_masks   records, for every fused activation the product applies (ops.conv1d / conv1d_windows / linear /
                  batch_norm with act != none), the mask (activation output > 0) under the name of the module that owns
                  the weight - the reference's own module names (state_dict prefixes);
  oracle_trace    runs oracle code and records the pre-activation tensor of every activation site under the same names
                  (the oracle's primitives take the state_dict prefix), in fp32 or fp64; with `impose` it REPLACES each
                  site's own sign test by the product's recorded mask: the oracle then evaluates the same piecewise-linear
                  function the product did, and its gradients can be compared element by element under strict bounds;
  audit           every element whose mask differs between product and fp32 oracle must have an fp64 pre-activation
                  within `tol` of the layer's scale from the kink (a wrong value would flip elements far from it).
"""
import contextlib

import torch
import torch.nn.functional as F

from music2dance_amd import ops
from oracle import m2d_oracle as O


class product_masks:
    """with product_masks(gen, critic) as pm: <product forward(s)>  ->  pm.masks[name] = [bool cpu tensors, call order]
    (name = the owning module's state_dict prefix, e.g. "decoder.blocks.0.bn2." or "stick_d.conv1.")"""

    def __init__(self, *modules):
        self.owner = {}
        for mod in modules:
            for name, p in mod.named_parameters():
                if name.endswith("weight"):
                    assert id(p) not in self.owner
                    self.owner[id(p)] = name[:-len("weight")]
        self.masks = {}

    def _rec(self, weight, mask):
        name = self.owner.get(id(weight))
        if name is not None:
            self.masks.setdefault(name, []).append(mask.detach().to("cpu"))

    def __enter__(self):
        self._orig = (ops.conv1d, ops.conv1d_windows, ops.linear, ops.batch_norm)
        o_conv, o_win, o_lin, o_bn = self._orig

        def conv1d(x, weight, bias=None, stride=1, padding=0, act=ops.ACT_NONE, *a, **k):
            out = o_conv(x, weight, bias, stride, padding, act, *a, **k)
            if act != ops.ACT_NONE:
                self._rec(weight, (out[0] if isinstance(out, tuple) else out) > 0)
            return out

        def conv1d_windows(track, T, hop, window, weight, bias=None, stride=1, padding=0, act=ops.ACT_NONE, *a, **k):
            out = o_win(track, T, hop, window, weight, bias, stride, padding, act, *a, **k)
            if act != ops.ACT_NONE:
                self._rec(weight, (out[0] if isinstance(out, tuple) else out) > 0)
            return out

        def linear(x, weight, bias=None, act=ops.ACT_NONE, slope=0.0):
            out = o_lin(x, weight, bias, act, slope)
            if act != ops.ACT_NONE:
                self._rec(weight, out > 0)
            return out

        def batch_norm(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, act=ops.ACT_NONE,
                       slope=0.0, residual=None, sums=None, out=None):
            y = o_bn(x, gamma, beta, running_mean, running_var, training, eps, momentum, act, slope, residual, sums, out)
            if act != ops.ACT_NONE:
                if residual is None:
                    self._rec(gamma, y > 0)
                else:
                    # y = residual + act(bn(x)): the mask is not recoverable from y (a tiny positive activation is lost
                    # in the sum) - evaluate the same kernels once more without the residual, on copies of the buffers
                    with torch.no_grad():
                        r = o_bn(x.detach(), gamma.detach(), beta.detach(), running_mean.clone(), running_var.clone(),
                                 training, eps, momentum, act, slope, None, sums, None)
                    self._rec(gamma, r > 0)
            return y

        ops.conv1d, ops.conv1d_windows, ops.linear, ops.batch_norm = conv1d, conv1d_windows, linear, batch_norm
        return self

    def __exit__(self, *exc):
        ops.conv1d, ops.conv1d_windows, ops.linear, ops.batch_norm = self._orig
        return False


class _FShim:
    """torch.nn.functional for the oracle module, with relu / leaky_relu observed (and optionally overridden)."""

    def __init__(self, trace):
        self._t = trace

    def __getattr__(self, name):
        return getattr(F, name)

    def relu(self, x, inplace=False):
        return self._t._activate(x, 0.0)

    def leaky_relu(self, x, negative_slope=0.01, inplace=False):
        return self._t._activate(x, float(negative_slope))


class oracle_trace:
    """with oracle_trace() as tr: <oracle calls>  ->  tr.sites[name] = [pre-activation tensors, call order]
    impose: {name: [bool masks]} (product_masks.masks) - every activation site uses ITS recorded mask (by call order)
    instead of its own sign test; a site without a recorded mask raises."""

    def __init__(self, impose=None, keep=True):
        self.impose = impose
        self.keep = keep
        self.sites = {}
        self._by_id = {}
        self._calls = {}

    def _wrap(self, fn):
        def wrapped(sd, prefix, x, *a, **k):
            out = fn(sd, prefix, x, *a, **k)
            self._by_id[id(out)] = (prefix, out)   # (holds a reference: the id stays unique while the trace lives)
            return out
        return wrapped

    def _activate(self, x, slope):
        ent = self._by_id.get(id(x))
        if ent is None:
            raise AssertionError("oracle_trace: an activation whose input is not the output of batch_norm / conv / linear")
        name = ent[0]
        n = self._calls.get(name, 0)
        self._calls[name] = n + 1
        if self.keep:
            self.sites.setdefault(name, []).append(x.detach())
        if self.impose is None:
            return F.leaky_relu(x, slope) if slope else F.relu(x)
        masks = self.impose.get(name)
        if not masks:
            raise AssertionError("oracle_trace: no product mask for activation site %r (call %d)" % (name, n))
        # (fewer product calls than oracle calls: the product evaluated a shared sub-network once - the critic's audio
        # branch, SURVEY.md A.6 - and audit() has checked that the oracle's repeated calls are bit-identical)
        m = masks[min(n, len(masks) - 1)].reshape(x.shape)
        return x * torch.where(m, torch.ones((), dtype=x.dtype), torch.full((), slope, dtype=x.dtype))

    def __enter__(self):
        self._orig = (O.batch_norm, O.conv, O.linear, O.F)
        O.batch_norm, O.conv, O.linear = self._wrap(O.batch_norm), self._wrap(O.conv), self._wrap(O.linear)
        O.F = _FShim(self)
        return self

    def __exit__(self, *exc):
        O.batch_norm, O.conv, O.linear, O.F = self._orig
        self._by_id = {}
        return False


def audit(masks, pre32, pre64, tol=2e-5, only=None):
    """masks: product_masks.masks; pre32 / pre64: oracle_trace.sites of the fp32 and the fp64 run of the same forward(s).
    -> (number of elements whose mask differs between the product and the fp32 oracle, worst |fp64 pre-activation| / scale
    among them, per-site counts). Asserts that the site lists match and that every differing element is a rounding flip."""
    total, worst, per = 0, 0.0, {}
    names = [n for n in pre32 if only is None or only(n)]
    for name in names:
        assert name in masks, "the product applied no fused activation for oracle site %r" % name
        nm = len(masks[name])
        assert 1 <= nm <= len(pre32[name]) == len(pre64[name]), \
            "%s: %d product calls, %d / %d oracle calls" % (name, nm, len(pre32[name]), len(pre64[name]))
        for i, (z32, z64) in enumerate(zip(pre32[name], pre64[name])):
            if i >= nm:
                # the product evaluated this site once where the oracle re-evaluates it on the same input
                assert torch.equal(z32, pre32[name][nm - 1]), "%s: oracle call %d is not a repeat of call %d" % (name, i, nm - 1)
                continue
            m = masks[name][i].reshape(z32.shape)
            d = m != (z32 > 0)
            n = int(d.sum())
            if n:
                scale = z64.abs().max().item()
                w = z64[d].abs().max().item() / scale
                assert w <= tol, "%s: an activation mask differs %.2e of the layer's scale away from the kink (> %.0e)" % (name, w, tol)
                worst = max(worst, w)
                per[name] = per.get(name, 0) + n
            total += n
    return total, worst, per
