"""Every kernel family repeated on the same inputs while a GEMM on another stream shares the CUs: bit-equal results every
time. In the training step the pose branch, the generator forward and the noise GRU run beside other launches all the time
(DESIGN.md 3.5); a kernel whose result depends on what else is resident is a parity failure no fixture of a lone launch can
see (round 6: csrc/tcn.hip, tests/test_gpu_tcn.py::test_results_do_not_depend_on_a_kernel_running_beside_them). Sizes are
small on purpose: a grid below 256 workgroups leaves CUs for the other stream's workgroups to land on.

Kernels that end in floating-point atomics (none of these forms) are not covered: they are equal to a few ulps only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPEATS = 60


def K():
    from music2dance_amd import kernels
    return kernels.impl()


def gen(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def conv_case(B, Cin, Cout, L, ks, stride, pad):
    x, w = gen(B, Cin, L, seed=1), gen(Cout, Cin, ks, seed=2, scale=1.0 / math.sqrt(Cin * ks))
    b = gen(Cout, seed=3, scale=0.1)
    Lo = (L + 2 * pad - ks) // stride + 1
    dy = gen(B, Cout, Lo, seed=4)
    return {
        "fwd": lambda: K().conv1d_fwd(x, w, b, stride, pad, act=2, slope=0.2),
        "fwd+stats": lambda: K().conv1d_fwd(x, w, b, stride, pad, with_stats=True),
        "bwd_data": lambda: K().conv1d_bwd_data(dy, w, L, stride, pad, dy_mask=dy, dy_mask_slope=0.2),
        "bwd_weight": lambda: K().conv1d_bwd_weight(x, dy, ks, stride, pad, with_bias=True),
    }


def cases():
    out = {}
    # the audio critic's k25 / stride-4 layers (tap-vectorised forward, sub-pixel backward-data), the encoder's k3 layers,
    # a strided U-Net-like layer, the single-channel first layer (thin kernels)
    for name, args in [("k25s4 64->128", (4, 64, 128, 1024, 25, 4, 11)), ("k25s4 16->32", (6, 16, 32, 2048, 25, 4, 11)),
                       ("k3 256->512", (40, 256, 512, 24, 3, 1, 1)), ("k4s2 32->64", (8, 32, 64, 500, 4, 2, 1)),
                       ("k25s4 1->32", (6, 1, 32, 4096, 25, 4, 11)), ("k25 69->128", (6, 69, 128, 120, 25, 1, 12)),
                       ("k250s50 1->32", (24, 1, 32, 3200, 250, 50, 124)), ("k160s4 1->32", (5, 1, 32, 3200, 160, 4, 79))]:
        for form, f in conv_case(*args).items():
            out["conv %s %s" % (name, form)] = f
    a, b, bias = gen(480, 256, seed=5), gen(256, 256, seed=6, scale=1 / 16.0), gen(256, seed=7)
    out["linear 480x256x256"] = lambda: K().gemm(0, a, b, bias, act=1)
    out["linear weight gradient"] = lambda: K().gemm(2, a, gen(480, 256, seed=8))
    a2, b2 = gen(6, 38400, seed=9), gen(100, 38400, seed=10, scale=0.01)
    out["head 6x100x38400"] = lambda: K().gemm(0, a2, b2)
    x = gen(480, 256, 1, seed=11)
    g, be = gen(256, seed=12), gen(256, seed=13)
    out["bn_stats rows"] = lambda: K().bn_stats(x)
    x3 = gen(6, 64, 2000, seed=14)
    out["bn_stats long"] = lambda: K().bn_stats(x3)
    out["channel_sums"] = lambda: K().channel_sums(x3)
    return out


@pytest.fixture(scope="module")
def busy():
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device=DEV)

    def go():
        with torch.cuda.stream(side):
            for _ in range(3):
                a @ a
    yield go
    torch.cuda.synchronize()


def _flat(o):
    return [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]


@pytest.mark.parametrize("name", sorted(cases().keys()) if torch.cuda.is_available() else [])
def test_bit_equal_beside_a_gemm_on_another_stream(name, busy):
    f = cases()[name]
    with K().weight_cache():
        ref = [t.clone() for t in _flat(f())]
        torch.cuda.synchronize()
        bad = 0
        for i in range(REPEATS):
            if i % 3 == 0:
                busy()
            got = _flat(f())
            bad += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
        torch.cuda.synchronize()
    assert bad == 0, "%d of %d repeats differ" % (bad, REPEATS)
