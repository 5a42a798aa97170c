"""utils.slice_audio_batch fused into the generator's first conv. The lazy
form returns the reference's slices as an in-place view of the padded track; the generator gives
the same poses and the same gradients from the view as from the materialised tensor."""
import pytest
import torch

from music2dance_amd import kernels
from music2dance_amd.layers import WindowView
from music2dance_amd.utils import slice_audio_batch


def test_lazy_slices_are_the_same_values_and_a_view():
    audio = torch.randn(3, 640 * 9, generator=torch.Generator().manual_seed(0))
    dense = slice_audio_batch(audio, 3200, 640, 2560)
    lazy = slice_audio_batch(audio, 3200, 640, 2560, lazy=True)
    assert dense.shape == lazy.shape == (3, 9, 3200) and torch.equal(dense, lazy)
    assert dense.is_contiguous() and lazy.stride() == (640 * 9 + 2560, 640, 1)
    wv = WindowView.of(lazy)
    assert wv is not None and (wv.T, wv.hop, wv.window) == (9, 640, 3200) and wv.track.shape == (3, 8 * 640 + 3200)
    assert WindowView.of(dense) is None  # a dense tensor has no overlapping windows to exploit
    assert torch.equal(wv.track.unfold(-1, 3200, 640), dense)


@pytest.mark.gpu
@pytest.mark.parametrize("enc", ["default", "unet", "wavegan"])
def test_generator_reads_windows_in_place(enc):
    from music2dance_amd.phase3.archis.default import SequenceGenerator
    assert kernels.impl().name == "hip"
    dev = torch.device("cuda:0")
    B, T = 3, 20
    torch.manual_seed(0)
    gen = SequenceGenerator(3200, 250, 250, 256, 69, 10, 2, 3, enc, "id", dev)
    audio = (0.1 * torch.randn(B, 640 * T, generator=torch.Generator().manual_seed(1))).to(dev)
    noise = torch.randn(B, T, 10, generator=torch.Generator().manual_seed(2)).to(dev)
    lazy = slice_audio_batch(audio, 3200, 640, 2560, lazy=True)
    dense = lazy.contiguous()
    sd0 = {k: v.clone() for k, v in gen.state_dict().items()}
    outs, grads = [], []
    for x in (dense, lazy):
        gen.load_state_dict(sd0)  # same BatchNorm buffers for both passes
        gen.train()
        gen.zero_grad(set_to_none=True)
        rows = gen(x, [T] * B, noise)
        rows.square().mean().backward()
        outs.append(rows.detach().clone())
        grads.append([p.grad.clone() for p in gen.parameters() if p.grad is not None])
    # Same kernels on the same values either way; the per-channel sums behind the BatchNorm statistics and the encoder's
    # bias gradients end in fp64 atomics whose order is not fixed, so a sum can differ in its last fp64 bits and - once in
    # a few hundred runs of this test, seen on cold boxes - round to the neighbouring fp32 value: equal to a few ulps, and
    # bit-equal everywhere else.
    def close(a, b):
        return torch.equal(a, b) or (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-30)
    assert close(outs[0], outs[1]), (outs[0] - outs[1]).abs().max()
    assert len(grads[0]) == len(grads[1])
    names = [n for n, p in gen.named_parameters() if p.grad is not None]
    off = [(n, (a - b).abs().max().item(), b.abs().max().item()) for n, a, b in zip(names, *grads) if not close(a, b)]
    assert not off, off


def test_draw_tape_replays_the_eager_host_draws_in_order():
    """layers.DrawTape (captured-graph mode of Phase1Engine): a body that draws noise, interpolation weights and a
    dropout mask from the host generator gets static buffers while the tape is installed; refill() then consumes the
    generator exactly as the eager body does - same values, same order - and a body that changes its draws is refused."""
    import torch
    from music2dance_amd import layers
    dev = torch.device("cpu")

    def body():
        a = layers.host_draw("randn", (4, 3), dev)
        b = layers.host_draw("rand", (4, 1), dev)
        c = layers.host_draw("bernoulli", (4, 5), dev, p=0.5)
        return a, b, c

    torch.manual_seed(5)
    eager = [t.clone() for t in body()] + [t.clone() for t in body()]  # two consecutive bodies
    tape = layers.DrawTape()
    with layers.draw_tape(tape):
        bufs = body()            # records; the buffers hold zeros
        tape.rewind()
        again = body()           # the capture pass gets the same buffers back
    assert all(x is y for x, y in zip(bufs, again)) and len(tape.entries) == 3
    assert all(float(t.abs().sum()) == 0.0 for t in bufs)
    torch.manual_seed(5)
    tape.refill()
    first = [t.clone() for t in bufs]
    tape.refill()
    second = [t.clone() for t in bufs]
    for got, want in zip(first + second, eager):
        assert torch.equal(got, want)
    # outside the tape the draws are ordinary host draws again
    torch.manual_seed(5)
    assert torch.equal(layers.host_draw("randn", (4, 3), dev), eager[0])
    # a body that draws something else than it recorded is an error, not a silent mismatch
    tape.rewind()
    with layers.draw_tape(tape):
        try:
            layers.host_draw("rand", (4, 3), dev)
            raise AssertionError("expected a RuntimeError")
        except RuntimeError:
            pass
