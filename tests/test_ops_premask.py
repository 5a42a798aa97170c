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


@pytest.mark.parametrize("training", [True, False])
def test_unet_block_without_a_graph_makes_its_concatenations_in_place(dev, training):
    """UBlock (phase3/archis/default.py:213-246 of the reference) under no_grad writes every skip concatenation's halves
    straight into one buffer (BatchNorm / upsampling with an output batch stride, max-pool reading a channel block):
    bit-identical to the torch.cat path the autograd forward takes, running statistics included."""
    import copy
    from music2dance_amd.phase3.archis.default import UBlock
    torch.manual_seed(3)
    blk = UBlock(16).to(dev)
    blk.train(training)
    x = torch.randn(5, 16, 200, generator=torch.Generator().manual_seed(1)).to(dev)
    ref_blk = copy.deepcopy(blk)
    ref = ref_blk(x.clone().requires_grad_(True)).detach()        # autograd path: torch.cat
    with torch.no_grad():
        got = blk(x)                                               # in-place concatenations
    assert torch.equal(got, ref)
    for a, b in zip(blk.buffers(), ref_blk.buffers()):
        assert torch.equal(a, b)
