"""The TemporalBlock kernels (csrc/tcn.hip: stride-1 "same" convolutions with 128 output channels, the pose critic of
phase3/archis/default.py:195-210 / phase2/archis/default.py:27-49 of the reference) against fp64 torch ops, through the
same C-ABI entry points as every other conv (m2d_conv1d_fwd / _fwd_sum / _bwd_data / _bwd_data_res / _bwd_weight_from):
the library routes the shapes that qualify to them, so these cases are sized to reach each tile width (96 / 64 / 32
columns), both sample-boundary situations (tiles inside one sample, tiles that straddle two), channel counts that leave
a ragged last 16-channel block, and every epilogue the critic iteration uses.

Tolerance: fp32 contraction noise over K = 7 * 128 terms, relative to the tensor's largest element: 2e-5 (3e-5 for the
weight gradient, whose K is the batch x length)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def K():
    from music2dance_amd import kernels
    return kernels.impl()


def rel_err(got, ref64):
    got = got.detach().cpu().double()
    ref64 = ref64.detach().cpu().double()
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    return (got - ref64).abs().max().item() / max(1.0, ref64.abs().max().item())


def gen(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# (B, Cin, L): tile width the launcher picks (the one-round plan with the least padded work, csrc/tcn.hip:
# m2d_tcn_conv_tile): 16 columns up to 4 096 positions, 32 up to 8 192, 48 up to 12 288, 64 up to 16 384, 96 beyond
SHAPES = [
    (3, 128, 120),      # 16-column tiles (16x16x4 MFMA blocks), three samples (the fixture size)
    (1, 128, 40),       # one sample shorter than two 32-column tiles
    (5, 128, 300),      # T = 300 (BASELINE configs[4]); tiles straddle at 300 = 18 x 16 + 12
    (7, 48, 44),        # three 16-channel blocks, L not a multiple of the tile
    (4, 100, 120),      # ragged last channel block (100 = 6 x 16 + 4)
    (50, 128, 120),     # 32-column tiles
    (32, 128, 120),     # a B-row tangent of BASELINE configs[1]: 240 tiles of 16
    (96, 128, 120),     # 3B rows of BASELINE configs[1]: 240 tiles of 48
    (130, 128, 120),    # 64-column tiles
    (140, 128, 120),    # 96-column tiles
    (192, 128, 120),    # 3B rows of the bench: exactly 240 tiles of 96
    (37, 128, 300),     # 48-column tiles at T = 300 (11 100 positions)
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_C%d_L%d" % s)
def test_temporal_conv_forward_and_backward_data(shape):
    B, Cin, L = shape
    x = gen(B, Cin, L, seed=1)
    w = gen(128, Cin, 7, seed=2, scale=1.0 / math.sqrt(Cin * 7))
    b = gen(128, seed=3, scale=0.1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=3)
    assert rel_err(K().conv1d_fwd(xd, wd, bd, 1, 3), ref) < 2e-5
    assert rel_err(K().conv1d_fwd(xd, wd, bd, 1, 3, act=1), ref.clamp_min(0)) < 2e-5
    # inside a packed-weight scope (the engines' way): the cached image is the one the kernel streams
    with K().weight_cache():
        assert rel_err(K().conv1d_fwd(xd, wd, None, 1, 3), ref - b.double().view(1, -1, 1)) < 2e-5
    if Cin == 128:
        # backward-data = the same kernel over the transposed image with the taps flipped
        dy = gen(B, 128, L, seed=4)
        x64 = x.double().requires_grad_(True)
        (gx,) = torch.autograd.grad(F.conv1d(x64, w.double(), None, padding=3), x64, dy.double())
        assert rel_err(K().conv1d_bwd_data(dy.to(DEV), wd, L, 1, 3), gx) < 2e-5


@pytest.mark.parametrize("shape", [(3, 128, 120), (60, 128, 120), (100, 128, 120), (130, 128, 120), (150, 128, 120)],
                         ids=lambda s: "B%d_C%d_L%d" % s)   # (16 / 32 / 48 / 64 / 96-column tiles)
def test_temporal_conv_epilogues(shape):
    """Every epilogue form of the critic iteration (critic_step.py): the block's second conv with two outputs, the
    tangent's mask-then-residual, backward-data with a masked dy, an output mask and the skip gradient (mask last)."""
    B, C, L = shape
    x = gen(B, C, L, seed=1)
    w = gen(128, C, 7, seed=2, scale=1.0 / math.sqrt(C * 7))
    b = gen(128, seed=3, scale=0.1)
    res, mask, om = gen(B, 128, L, seed=5), gen(B, 128, L, seed=6), gen(B, C, L, seed=7)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=3)
    slope = 0.2
    m64 = torch.where(mask.double() > 0, torch.ones_like(ref), torch.full_like(ref, slope))
    # leaky + out mask + residual: mask BEFORE the residual
    y = K().conv1d_fwd(xd, wd, bd, 1, 3, act=2, slope=slope, residual=res.to(DEV), out_mask=mask.to(DEV), out_mask_slope=slope)
    assert rel_err(y, F.leaky_relu(ref, slope) * m64 + res.double()) < 2e-5
    # two outputs into views of larger buffers (the 3B-row buffers of the critic iteration)
    big = torch.zeros(B + 2, 128, L, device=DEV)
    big2 = torch.zeros(B + 1, 128, L, device=DEV)
    y2, so = K().conv1d_fwd(xd, wd, bd, 1, 3, act=1, residual=res.to(DEV), out=big[1:B + 1], sum_out=big2[1:])
    assert float(big[0].abs().max()) == 0 and float(big[B + 1].abs().max()) == 0 and float(big2[0].abs().max()) == 0
    assert rel_err(y2, F.relu(ref)) < 2e-5 and rel_err(so, F.relu(ref) + res.double()) < 2e-5
    # in-place masking: out aliases out_mask (the tangent pass writes over the rows whose masks it reads)
    buf = mask.to(DEV).clone()
    K().conv1d_fwd(xd, wd, None, 1, 3, out_mask=buf, out_mask_slope=0.0, out=buf)
    assert rel_err(buf, F.conv1d(x.double(), w.double(), None, padding=3) * (mask.double() > 0)) < 2e-5
    # backward-data: masked dy, output mask with slope, skip gradient added before the mask
    dy = gen(B, 128, L, seed=4)
    m0 = (mask.double() > 0).double()
    x64 = x.double().requires_grad_(True)
    out = F.conv1d(x64, w.double(), None, padding=3)
    (gx,) = torch.autograd.grad(out, x64, dy.double() * m0)
    omf = torch.where(om.double() > 0, torch.ones((), dtype=torch.float64), torch.full((), 0.2, dtype=torch.float64))
    rs = gen(B, C, L, seed=8)
    dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, 1, 3, dy_mask=mask.to(DEV), dy_mask_slope=0.0, out_mask=om.to(DEV),
                             out_mask_slope=0.2, residual=rs.to(DEV))
    assert rel_err(dx, (gx + rs.double()) * omf) < 2e-5
    dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, 1, 3, dy_mask=mask.to(DEV), dy_mask_slope=0.0)
    assert rel_err(dx, gx) < 2e-5


@pytest.mark.parametrize("shape", [(3, 128, 120), (6, 128, 300), (9, 64, 44), (5, 64, 120), (4, 48, 60), (2, 100, 120), (96, 128, 120),
                                   (192, 128, 120)],
                         ids=lambda s: "B%d_C%d_L%d" % s)
def test_temporal_conv_weight_gradient(shape):
    B, C, L = shape
    x = gen(B, C, L, seed=1)
    dy = gen(B, 128, L, seed=4)
    mask = gen(B, 128, L, seed=6)
    m0 = (mask.double() > 0).double()
    w64 = torch.zeros(128, C, 7, dtype=torch.float64, requires_grad=True)
    out = F.conv1d(x.double(), w64, None, padding=3)
    (gw,) = torch.autograd.grad(out, w64, dy.double(), retain_graph=True)
    (gwm,) = torch.autograd.grad(out, w64, dy.double() * m0)
    xd, dyd = x.to(DEV), dy.to(DEV)
    assert rel_err(K().conv1d_bwd_weight(xd, dyd, 7, 1, 3), gw) < 3e-5
    dw, db = K().conv1d_bwd_weight(xd, dyd, 7, 1, 3, with_bias=True)
    assert rel_err(dw, gw) < 3e-5 and rel_err(db, dy.double().sum((0, 2))) < 3e-5
    dw, db = K().conv1d_bwd_weight(xd, dyd, 7, 1, 3, dy_mask=mask.to(DEV), dy_mask_slope=0.0, with_bias=True)
    assert rel_err(dw, gwm) < 3e-5 and rel_err(db, (dy.double() * m0).sum((0, 2))) < 3e-5
    if B >= 3:
        # the bias gradient over the samples [B // 3, B) only: the first rows pair second-order operands
        f = B // 3
        dw, db = K().conv1d_bwd_weight(xd, dyd, 7, 1, 3, dy_mask=mask.to(DEV), dy_mask_slope=0.0, with_bias=True,
                                       bias_from_sample=f)
        assert rel_err(dw, gwm) < 3e-5 and rel_err(db, (dy.double() * m0)[f:].sum((0, 2))) < 3e-5
    # deterministic: the partial tiles are summed in a fixed order
    a = K().conv1d_bwd_weight(xd, dyd, 7, 1, 3)
    bb = K().conv1d_bwd_weight(xd, dyd, 7, 1, 3)
    assert torch.equal(a, bb)


def test_results_do_not_depend_on_a_kernel_running_beside_them():
    """A small launch (45 workgroups of 16 columns) repeated while a GEMM on another stream shares its CUs: bit-equal every
    time. Round 6: the read-ahead past the last k-step landed, under LDS contention, in registers the compiler had already
    reused for the epilogue's addresses - wrong 16 x 12 blocks in one launch of five, never when alone on the chip
    (tools/tcn_determinism.py is the long form over every tile width)."""
    B, L = 6, 120
    x, dy = gen(B, 128, L, seed=11).to(DEV), gen(B, 128, L, seed=12).to(DEV)
    w = gen(128, 128, 7, seed=13, scale=1.0 / math.sqrt(128 * 7)).to(DEV)
    b = gen(128, seed=14, scale=0.1).to(DEV)
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device=DEV)
    forms = [lambda: K().conv1d_fwd(x, w, b, 1, 3, act=1), lambda: K().conv1d_bwd_data(dy, w, L, 1, 3)]
    with K().weight_cache():
        refs = [f().clone() for f in forms]
        torch.cuda.synchronize()
        bad = [0, 0]
        for i in range(150):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a @ a
            for k, f in enumerate(forms):
                bad[k] += int(not torch.equal(f(), refs[k]))
        torch.cuda.synchronize()
    assert bad == [0, 0], bad
