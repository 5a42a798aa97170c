"""Pre-masked gradient contract of ops.conv1d (in_act / out_pm): a chain of fused conv +
activation layers gives the same first- and second-order gradients with the flags (masks applied
once, in the consumer's backward-data epilogue) as without them (masked operand loads), for
ReLU and for LeakyReLU (whose mask is not idempotent, so a double application would show)."""
import pytest
import torch

from music2dance_amd import kernels, ops


@pytest.fixture(params=[pytest.param("cpu-fake"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    if request.param == "hip":
        yield torch.device("cuda:0")
    else:
        from tests.fake_backend import FakeKernels
        prev = kernels.set_impl(FakeKernels())
        try:
            yield torch.device("cpu")
        finally:
            kernels.set_impl(prev)


def _chain(x, ws, bs, act, slope, flags):
    n = len(ws)
    for i, (w, b) in enumerate(zip(ws, bs)):
        last = i == n - 1
        a = ops.ACT_NONE if last else act
        kw = {}
        if flags:
            kw = dict(in_act=(act, slope) if i > 0 else None, out_pm=not last)
        x = ops.conv1d(x, w, b, stride=2 if i == 1 else 1, padding=1, act=a, slope=slope, **kw)
    return x


@pytest.mark.parametrize("act,slope", [(ops.ACT_RELU, 0.0), (ops.ACT_LEAKY, 0.2)])
def test_premasked_chain_matches_masked_operands(dev, act, slope):
    g = torch.Generator().manual_seed(0)
    chans = [3, 16, 24, 16, 5]
    ws = [(torch.randn(chans[i + 1], chans[i], 3, generator=g) * 0.4).to(dev).requires_grad_(True) for i in range(4)]
    bs = [(torch.randn(chans[i + 1], generator=g) * 0.1).to(dev).requires_grad_(True) for i in range(4)]
    x0 = torch.randn(4, 3, 40, generator=g).to(dev)
    res = []
    for flags in (False, True):
        x = x0.clone().requires_grad_(True)
        y = _chain(x, ws, bs, act, slope, flags)
        score = y.sum((1, 2))
        # gradient penalty form: first backward w.r.t. the input with a graph, then backward of its norm
        (gx,) = torch.autograd.grad(score, x, torch.ones_like(score), create_graph=True)
        pen = ((gx.reshape(4, -1).norm(dim=1) - 1) ** 2).mean()
        loss = pen + 0.1 * (y ** 2).mean()
        grads = torch.autograd.grad(loss, ws + bs)
        res.append([y.detach(), gx.detach()] + [t.detach() for t in grads])
    for a, b in zip(*res):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7
