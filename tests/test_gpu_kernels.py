"""Op-level parity of every HIP kernel (through the C-ABI) against plain fp64/fp32 torch
CPU ops — the primitives the oracle (oracle/) is written in. Covers every
(Cin, Cout, k, stride, pad, L) the hot path uses at small batch.

Tolerance: fp32 contraction noise. err = max|got - ref64| / max(1, max|ref64|) <= 2e-5
for contractions (K up to 38 400 terms), 1e-6-level for elementwise kernels.
"""
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
    denom = max(1.0, ref64.abs().max().item())
    return (got - ref64).abs().max().item() / denom


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# (name, B, Cin, L, Cout, k, stride, pad)
CONV_CASES = [
    ("stick.conv1", 3, 69, 120, 128, 25, 1, 12),
    ("stick.conv1.k9", 2, 69, 120, 128, 9, 1, 4),
    ("temporal.k7", 3, 128, 120, 128, 7, 1, 3),
    ("stick.fconv", 5, 128, 120, 100, 120, 1, 0),
    ("p2.lastconv", 4, 128, 120, 1, 120, 1, 0),
    ("audio_d.l1", 2, 1, 76800, 32, 25, 4, 11),
    ("audio_d.l2", 2, 32, 19200, 64, 25, 4, 11),
    ("audio_d.l3", 2, 64, 4800, 128, 25, 4, 11),
    ("audio_d.l4", 2, 128, 1200, 256, 25, 4, 11),
    ("audio_d.l5", 2, 256, 300, 512, 25, 4, 11),
    ("audio_d.l6", 3, 512, 75, 100, 75, 1, 0),
    ("enc.c0", 6, 1, 3200, 32, 250, 50, 124),
    ("enc.c1", 6, 32, 64, 64, 4, 2, 1),
    ("enc.c2", 6, 64, 32, 128, 4, 2, 1),
    ("enc.c3", 6, 128, 16, 256, 4, 2, 1),
    ("enc.c4", 6, 256, 8, 512, 4, 2, 1),
    ("enc.c5", 6, 512, 4, 1024, 4, 2, 1),
    ("enc.c6", 6, 1024, 2, 250, 2, 1, 0),
    ("wg.l1", 3, 1, 3200, 32, 25, 4, 0),
    ("wg.l1-odd-lout", 3, 1, 3204, 32, 25, 4, 0),
    ("wg.l1-pad-even", 2, 1, 3210, 32, 25, 4, 3),
    ("wg.l2", 3, 32, 794, 64, 25, 4, 0),
    ("wg.l3", 3, 64, 193, 128, 25, 4, 0),
    ("wg.l4", 3, 128, 43, 256, 25, 4, 0),
    ("wg.l5", 3, 256, 5, 250, 5, 1, 0),
    ("unet.c0", 2, 1, 3200, 32, 160, 4, 79),
    ("unet.c1", 2, 32, 800, 64, 4, 2, 1),
    ("unet.k3", 2, 128, 200, 128, 3, 1, 1),
    ("unet.k3cat", 2, 256, 50, 128, 3, 1, 1),
    ("unet.k3.25", 2, 128, 25, 128, 3, 1, 1),
    ("unet.fc", 3, 128, 200, 250, 200, 1, 0),
    ("odd.a", 2, 3, 17, 5, 3, 2, 1),
    ("odd.b", 1, 7, 33, 9, 5, 3, 0),
    ("odd.c", 2, 2, 40, 33, 6, 4, 5),
    # channel counts >= 16 that are not multiples of the 16-deep chunk (lo tails of the packed path)
    ("odd.d", 2, 20, 50, 24, 5, 3, 2),
    ("odd.e", 3, 17, 21, 40, 2, 2, 0),
    ("odd.f", 2, 33, 64, 70, 7, 1, 3),
    # batch large enough for split-K plans and several column tiles per sample boundary
    ("split.a", 40, 32, 100, 48, 5, 2, 2),
    ("split.b", 70, 16, 33, 130, 3, 1, 1),
    # tap-vectorised k25 / s4 / pad 11 forwards in the phantom-paired K order: channel counts 16 / 48 /
    # 64 and batches whose split-K ranges begin inside each of its three regions (full groups, first groups, last groups)
    ("k4pair.a", 1, 16, 512, 64, 25, 4, 11),
    ("k4pair.b", 3, 64, 4800, 128, 25, 4, 11),
    ("k4pair.c", 5, 48, 1024, 64, 25, 4, 11),
    ("k4pair.d", 9, 32, 2048, 64, 25, 4, 11),
    # the same kernel on a layer it walks in the plain order (k24: no partial groups at the end)
    ("k4plain.a", 2, 32, 1024, 64, 24, 4, 11),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv1d_fwd_bwd(case):
    name, B, Cin, L, Cout, ks, s, p = case
    x = gen(B, Cin, L, seed=1)
    w = gen(Cout, Cin, ks, seed=2, scale=1.0 / math.sqrt(Cin * ks))
    b = gen(Cout, seed=3, scale=0.1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = F.conv1d(x.double(), w.double(), b.double(), stride=s, padding=p)
    y = K().conv1d_fwd(xd, wd, bd, s, p)
    assert rel_err(y, ref) < 2e-5, "fwd"
    # fused bias + ReLU
    y = K().conv1d_fwd(xd, wd, bd, s, p, act=1)
    assert rel_err(y, ref.clamp_min(0)) < 2e-5, "fwd+relu"
    # backward-data / backward-weight vs autograd of the fp64 op
    dy = gen(*ref.shape, seed=4)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    out = F.conv1d(x64, w64, None, stride=s, padding=p)
    gx, gw = torch.autograd.grad(out, (x64, w64), dy.double())
    dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p)
    assert rel_err(dx, gx) < 2e-5, "bwd_data"
    dw = K().conv1d_bwd_weight(xd, dy.to(DEV), ks, s, p)
    assert rel_err(dw, gw) < 3e-5, "bwd_weight"
    # bias gradient from the same launch (all-ones operand column / spare MFMA row)
    dw2, db = K().conv1d_bwd_weight(xd, dy.to(DEV), ks, s, p, with_bias=True)
    assert rel_err(dw2, gw) < 3e-5, "bwd_weight (with bias)"
    assert rel_err(db, dy.double().sum((0, 2))) < 3e-5, "bias gradient"


def test_conv1d_cabi_without_packed_weights():
    """Straight through the C-ABI with w_packed = NULL: the call packs into its workspace."""
    import ctypes
    from music2dance_amd import _lib
    h = _lib.lib()
    B, Cin, L, Cout, ks, s, p = 3, 32, 40, 24, 5, 2, 2
    x, w = gen(B, Cin, L, seed=1).to(DEV), gen(Cout, Cin, ks, seed=2, scale=0.1).to(DEV)
    Lout = (L + 2 * p - ks) // s + 1
    y = torch.empty(B, Cout, Lout, device=DEV)
    nws = h.m2d_conv1d_workspace_bytes(0, B, Cin, L, Cout, ks, s, p)
    assert nws >= Cout * Cin * ks * 4
    ws = torch.empty(nws // 4 + 1, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = h.m2d_conv1d_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, Cin, L, Cout, ks, s, p, 0, 0.0,
                          None, None, 0.0, None, ws.data_ptr(), nws, st)
    assert rc == 0, h.m2d_last_error()
    ref = F.conv1d(x.cpu().double(), w.cpu().double(), None, stride=s, padding=p)
    assert rel_err(y, ref) < 2e-5
    dy = gen(B, Cout, Lout, seed=3).to(DEV)
    dx = torch.empty(B, Cin, L, device=DEV)
    nws = h.m2d_conv1d_workspace_bytes(1, B, Cin, L, Cout, ks, s, p)
    ws = torch.empty(nws // 4 + 1, device=DEV)
    rc = h.m2d_conv1d_bwd_data(dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), B, Cin, L, Cout, ks, s, p, None, 0.0,
                               None, 0.0, ws.data_ptr(), nws, st)
    assert rc == 0, h.m2d_last_error()
    x64 = x.cpu().double().requires_grad_(True)
    (gx,) = torch.autograd.grad(F.conv1d(x64, w.cpu().double(), None, stride=s, padding=p), x64, dy.cpu().double())
    assert rel_err(dx, gx) < 2e-5
    # too small a workspace is refused, not overrun
    rc = h.m2d_conv1d_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, Cin, L, Cout, ks, s, p, 0, 0.0,
                          None, None, 0.0, None, ws.data_ptr(), 16, st)
    assert rc != 0


def test_packed_weight_cache_scope():
    """Outside weight_cache() every call packs afresh; inside, the images are reused until
    invalidate_packed() - in-place updates that do not move the version counter (fused optimizers)
    are only seen after it."""
    k = K()
    x, w = gen(2, 32, 30, seed=1).to(DEV), gen(16, 32, 3, seed=2, scale=0.2).to(DEV)
    ref = lambda: F.conv1d(x.cpu().double(), w.cpu().double(), None, padding=1)
    assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5
    w.mul_(2.0)
    assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5          # no scope: fresh
    with k.weight_cache():
        n0 = k.pack_launches
        assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5
        assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5
        assert k.pack_launches == n0 + 1                                   # packed once
        torch._foreach_mul_([w], 0.5)                                      # version-less style update ...
        k.invalidate_packed()                                              # ... announced by the caller
        assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5
        w.add_(1.0)                                                        # ordinary in-place op: version moves
        assert rel_err(k.conv1d_fwd(x, w, None, 1, 1), ref()) < 2e-5
    assert not k._packed


@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[2], CONV_CASES[6], CONV_CASES[12], CONV_CASES[29]],
                         ids=lambda c: c[0])
def test_conv1d_masks_and_epilogue(case):
    name, B, Cin, L, Cout, ks, s, p = case
    x = gen(B, Cin, L, seed=1)
    w = gen(Cout, Cin, ks, seed=2, scale=1.0 / math.sqrt(Cin * ks))
    b = gen(Cout, seed=3, scale=0.1)
    ref = F.conv1d(x.double(), w.double(), b.double(), stride=s, padding=p)
    mask = gen(*ref.shape, seed=5)
    res = gen(*ref.shape, seed=6)
    slope = 0.2
    m64 = torch.where(mask.double() > 0, torch.ones_like(ref), torch.full_like(ref, slope))
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    # leaky + residual + out mask
    y = K().conv1d_fwd(xd, wd, bd, s, p, act=2, slope=slope, residual=res.to(DEV), out_mask=mask.to(DEV),
                       out_mask_slope=slope)
    want = F.leaky_relu(ref, slope) * m64 + res.double()  # mask BEFORE the residual (include/m2d.h)
    assert rel_err(y, want) < 2e-5
    # two outputs: y without the residual (the backward's activation mask), sum_out = y + residual; y into a
    # caller-provided view of a larger buffer
    big = torch.zeros((B + 2,) + tuple(ref.shape[1:]), device=DEV)
    y2, so = K().conv1d_fwd(xd, wd, bd, s, p, act=1, residual=res.to(DEV), out=big[1:B + 1],
                            sum_out=torch.empty(tuple(ref.shape), device=DEV))
    assert y2.data_ptr() == big[1:].data_ptr() and float(big[0].abs().max()) == 0 and float(big[B + 1].abs().max()) == 0
    assert rel_err(y2, F.relu(ref)) < 2e-5 and rel_err(so, F.relu(ref) + res.double()) < 2e-5
    # in-place masking: out aliases out_mask
    buf = mask.to(DEV).clone()
    K().conv1d_fwd(xd, wd, None, s, p, out_mask=buf, out_mask_slope=0.0, out=buf)
    assert rel_err(buf, F.conv1d(x.double(), w.double(), None, stride=s, padding=p) * (mask.double() > 0)) < 2e-5
    # masked dy in both backward halves (mask slope 0 = ReLU derivative)
    dy = gen(*ref.shape, seed=4)
    m0 = (mask.double() > 0).double()
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    out = F.conv1d(x64, w64, None, stride=s, padding=p)
    gx, gw = torch.autograd.grad(out, (x64, w64), dy.double() * m0)
    dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p, dy_mask=mask.to(DEV), dy_mask_slope=0.0)
    assert rel_err(dx, gx) < 2e-5
    dw, db = K().conv1d_bwd_weight(xd, dy.to(DEV), ks, s, p, dy_mask=mask.to(DEV), dy_mask_slope=0.0, with_bias=True)
    assert rel_err(dw, gw) < 3e-5
    assert rel_err(db, (dy.double() * m0).sum((0, 2))) < 3e-5
    # epilogue mask on backward-data (shape of dx), with and without the operand mask, slope 0.2
    if not (Cin == 1 and ks == 25 and s == 4):
        om = gen(B, Cin, L, seed=6)
        omf = torch.where(om.double() > 0, torch.ones((), dtype=torch.float64), torch.full((), 0.2, dtype=torch.float64))
        dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p, dy_mask=mask.to(DEV), dy_mask_slope=0.0,
                                 out_mask=om.to(DEV), out_mask_slope=0.2)
        assert rel_err(dx, gx * omf) < 2e-5
        (gx_plain,) = torch.autograd.grad(F.conv1d(x64, w64, None, stride=s, padding=p), x64, dy.double())
        dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p, out_mask=om.to(DEV), out_mask_slope=0.2)
        assert rel_err(dx, gx_plain * omf) < 2e-5
        # skip-connection gradient added in the epilogue, BEFORE the mask: dx = mask * (conv^T(dy) + residual)
        rs = gen(B, Cin, L, seed=8)
        dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p, out_mask=om.to(DEV), out_mask_slope=0.2, residual=rs.to(DEV))
        assert rel_err(dx, (gx_plain + rs.double()) * omf) < 2e-5
        dx = K().conv1d_bwd_data(dy.to(DEV), wd, L, s, p, residual=rs.to(DEV))
        assert rel_err(dx, gx_plain + rs.double()) < 2e-5
        # bias gradient over the samples [1, B) only (rows in front pair second-order operands)
        if ref.shape[2] >= 16 and B > 1:
            dw2, db2 = K().conv1d_bwd_weight(xd, dy.to(DEV), ks, s, p, dy_mask=mask.to(DEV), dy_mask_slope=0.0,
                                             with_bias=True, bias_from_sample=1)
            assert rel_err(dw2, gw) < 3e-5
            assert rel_err(db2, (dy.double() * m0)[1:].sum((0, 2))) < 3e-5


GEMM_CASES = [(64, 128, 200), (7680, 256, 250), (3840, 720, 250), (33, 69, 256), (1, 1, 128), (100, 1, 128),
              (257, 131, 77), (64, 15360, 100)]


@pytest.mark.parametrize("mnk", GEMM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_gemm_modes(mnk):
    M, N, Kd = mnk
    a = gen(M, Kd, seed=1)
    bt = gen(N, Kd, seed=2, scale=1.0 / math.sqrt(Kd))
    bias = gen(N, seed=3)
    ref = a.double() @ bt.double().t() + bias.double()
    c = K().gemm(0, a.to(DEV), bt.to(DEV), bias.to(DEV))
    assert rel_err(c, ref) < 2e-5, "NT"
    c = K().gemm(0, a.to(DEV), bt.to(DEV), bias.to(DEV), act=1)
    assert rel_err(c, ref.clamp_min(0)) < 2e-5, "NT+relu"
    b = bt.t().contiguous()  # (K, N)
    c = K().gemm(1, a.to(DEV), b.to(DEV))
    assert rel_err(c, a.double() @ b.double()) < 2e-5, "NN"
    at = a.t().contiguous()  # (K, M)
    c = K().gemm(2, at.to(DEV), b.to(DEV))
    assert rel_err(c, at.double().t() @ b.double()) < 2e-5, "TN"
    # masks
    am = gen(M, Kd, seed=7)
    om = gen(M, N, seed=8)
    a_eff = a.double() * torch.where(am.double() > 0, 1.0, 0.0)
    want = (a_eff @ b.double()) * torch.where(om.double() > 0, 1.0, 0.2)
    c = K().gemm(1, a.to(DEV), b.to(DEV), a_mask=am.to(DEV), a_mask_slope=0.0, out_mask=om.to(DEV),
                 out_mask_slope=0.2)
    assert rel_err(c, want) < 2e-5, "NN masked"


def test_gemm_tn_long_k():
    # dW = dy^T x with K = B*T rows (split-K path)
    Kd, M, N = 7680, 256, 250
    a = gen(Kd, M, seed=1)
    b = gen(Kd, N, seed=2)
    c = K().gemm(2, a.to(DEV), b.to(DEV))
    assert rel_err(c, a.double().t() @ b.double()) < 3e-5


BN_CASES = [(64, 128, 1), (7680, 256, 1), (48, 32, 64), (48, 1024, 2), (24, 64, 193), (24, 128, 43), (24, 256, 5),
            (6, 32, 800), (6, 128, 25), (5, 3, 7)]


@pytest.mark.parametrize("shape", BN_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bn_train_fwd_bwd(shape, act):
    B, C, L = shape
    x = gen(B, C, L, seed=1) * 1.5 + 0.3
    if L == 1:
        x = x.view(B, C)
    gamma = 1.0 + gen(C, seed=2, scale=0.1)
    beta = gen(C, seed=3, scale=0.1)
    rm = gen(C, seed=4, scale=0.1)
    rv = 1.0 + gen(C, seed=5, scale=0.1).abs()
    slope = 0.2
    x64 = x.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = beta.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    ref = z if act == 0 else (F.relu(z) if act == 1 else F.leaky_relu(z, slope))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y, mean, invstd = K().bn_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), rmd, rvd, True, 1e-5, 0.1, act, slope)
    assert rel_err(y, ref) < 1e-5
    assert rel_err(rmd, rm64) < 1e-6 and rel_err(rvd, rv64) < 1e-6
    dy = gen(*x.shape, seed=6)
    gx, gg, gb = torch.autograd.grad(ref, (x64, g64, b64), dy.double())
    dx, dg, db = K().bn_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), beta.to(DEV), mean, invstd, act, slope)
    assert rel_err(dx, gx) < 2e-5
    assert rel_err(dg, gg) < 2e-5 and rel_err(db, gb) < 2e-5


def test_bn_eval_and_residual():
    B, C = 40, 256
    x = gen(B, C, seed=1)
    res = gen(B, C, seed=9)
    gamma, beta = 1.0 + gen(C, seed=2, scale=0.1), gen(C, seed=3, scale=0.1)
    rm, rv = gen(C, seed=4, scale=0.1), 1.0 + gen(C, seed=5, scale=0.1).abs()
    ref = F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y, _, _ = K().bn_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), rmd, rvd, False, 1e-5, 0.1, 1, 0.0,
                         residual=res.to(DEV))
    assert rel_err(y, res.double() + F.relu(ref)) < 1e-5
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)


@pytest.mark.parametrize("shape", [(4, 128, 120), (7, 5, 3), (64, 100, 1), (3, 32, 19200), (2, 5000, 3)],   # (5 000 > the fixed scratch: stateless form)
                         ids=lambda c: "x".join(map(str, c)))
def test_channel_sums(shape):
    x = gen(*shape, seed=1)
    m = gen(*shape, seed=2)
    got = K().channel_sums(x.to(DEV))
    assert rel_err(got, x.double().sum(dim=(0, 2))) < 1e-5
    got = K().channel_sums(x.to(DEV), m.to(DEV), 0.2)
    want = (x.double() * torch.where(m.double() > 0, 1.0, 0.2)).sum(dim=(0, 2))
    assert rel_err(got, want) < 1e-5


def _gru_ref(x, w_ih, w_hh, b_ih, b_hh, lengths=None):
    I, H = w_ih.shape[1], w_hh.shape[1]
    rnn = torch.nn.GRU(I, H, 1, batch_first=True).double()
    with torch.no_grad():
        rnn.weight_ih_l0.copy_(w_ih)
        rnn.weight_hh_l0.copy_(w_hh)
        rnn.bias_ih_l0.copy_(b_ih)
        rnn.bias_hh_l0.copy_(b_hh)
    return rnn


@pytest.mark.parametrize("dims", [(5, 7, 250, 240), (3, 12, 10, 10), (32, 9, 50, 50), (17, 5, 33, 21)],
                         ids=lambda c: "x".join(map(str, c)))
def test_gru_layer_fwd_bwd(dims):
    B, T, I, H = dims
    x = gen(B, T, I, seed=1)
    w_ih = gen(3 * H, I, seed=2, scale=1 / math.sqrt(I))
    w_hh = gen(3 * H, H, seed=3, scale=1 / math.sqrt(H))
    b_ih, b_hh = gen(3 * H, seed=4, scale=0.1), gen(3 * H, seed=5, scale=0.1)
    rnn = _gru_ref(x, w_ih, w_hh, b_ih, b_hh)
    x64 = x.double().requires_grad_(True)
    ref, _ = rnn(x64)
    k = K()
    gi = k.gemm(0, x.view(B * T, I).to(DEV), w_ih.to(DEV), b_ih.to(DEV)).view(B, T, 3 * H)
    out, saved = k.gru_layer_fwd(gi, w_hh.t().contiguous().to(DEV), b_hh.to(DEV))
    assert rel_err(out, ref) < 1e-5
    dout = gen(B, T, H, seed=6)
    grads = torch.autograd.grad(ref, (x64, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0),
                                dout.double())
    dgi, dgh = k.gru_layer_bwd(dout.to(DEV), out, saved, w_hh.to(DEV))
    dgi2 = dgi.view(B * T, 3 * H)
    dgh2 = dgh.view(B * T, 3 * H)
    dx = k.gemm(1, dgi2, w_ih.to(DEV)).view(B, T, I)
    dw_ih = k.gemm(2, dgi2, x.view(B * T, I).to(DEV))
    hprev = torch.cat([torch.zeros(B, 1, H, device=DEV), out[:, :-1]], 1).contiguous().view(B * T, H)
    dw_hh = k.gemm(2, dgh2, hprev)
    assert rel_err(dx, grads[0]) < 2e-5
    assert rel_err(dw_ih, grads[1]) < 2e-5
    assert rel_err(dw_hh, grads[2]) < 2e-5
    assert rel_err(k.channel_sums(dgi2.view(B * T, 3 * H, 1)), grads[3]) < 2e-5
    assert rel_err(k.channel_sums(dgh2.view(B * T, 3 * H, 1)), grads[4]) < 2e-5


def test_gru_lengths():
    B, T, I, H = 4, 6, 8, 16
    lengths = [6, 5, 3, 1]
    x = gen(B, T, I, seed=1)
    w_ih, w_hh = gen(3 * H, I, seed=2, scale=0.3), gen(3 * H, H, seed=3, scale=0.3)
    b_ih, b_hh = gen(3 * H, seed=4, scale=0.1), gen(3 * H, seed=5, scale=0.1)
    rnn = _gru_ref(x, w_ih, w_hh, b_ih, b_hh)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x.double(), lengths, batch_first=True)
    ref, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True)
    k = K()
    gi = k.gemm(0, x.view(B * T, I).to(DEV), w_ih.to(DEV), b_ih.to(DEV)).view(B, T, 3 * H)
    lens = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    out, _ = k.gru_layer_fwd(gi, w_hh.t().contiguous().to(DEV), b_hh.to(DEV), lengths=lens)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("n", [69, 8280, 76800])
@pytest.mark.parametrize("lp", [False, True])
def test_gp_ops(n, lp):
    B = 6
    real, fake = gen(B, n, seed=1), gen(B, n, seed=2)
    alpha = torch.rand(B, generator=torch.Generator().manual_seed(3))
    got = K().gp_interpolate(real.to(DEV), fake.to(DEV), alpha.to(DEV))
    want = alpha.view(B, 1) * real + (1 - alpha.view(B, 1)) * fake  # fp32, same three roundings
    assert torch.equal(got.cpu(), want)
    scale = torch.tensor([0.2, 0.9, 1.0, 1.1, 3.0, 0.0]).view(B, 1) / math.sqrt(n)
    g = (gen(B, n, seed=4) * scale)
    g64 = g.double().requires_grad_(True)
    if lp:
        d = (g64.norm(2, dim=1) - 1).clamp_min(0)
        ref = (d ** 2).mean()
    else:
        ref = ((torch.sqrt((g64 ** 2).sum(1) + 1e-12) - 1) ** 2).mean()
    pen, norms = K().gp_penalty_fwd(g.to(DEV), lp)
    assert abs(pen.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    (gref,) = torch.autograd.grad(ref, g64)
    gout = torch.tensor(0.7, device=DEV)
    dg = K().gp_penalty_bwd(g.to(DEV), norms, gout, lp)
    err = (dg.cpu().double() - 0.7 * gref).abs().max().item()
    assert err < 1e-5 * max(1e-3, gref.abs().max().item()) + 1e-9


def test_l1_tv():
    B, C, T = 5, 69, 120
    a, b = gen(B, C, T, seed=1), gen(B, C, T, seed=2)
    a64 = a.double().requires_grad_(True)
    ref = (a64 - b.double()).abs().mean()
    got = K().l1_mean_fwd(a.to(DEV), b.to(DEV))
    assert abs(got.item() - ref.item()) < 1e-6
    (gref,) = torch.autograd.grad(ref, a64)
    gout = torch.tensor(1.3, device=DEV)
    da = K().l1_mean_bwd(a.to(DEV), b.to(DEV), gout)
    assert rel_err(da, 1.3 * gref) < 1e-6
    # TV on the generator's native (B*T, C) layout, viewed as (B, C, T)
    rows = gen(B * T, C, seed=3)
    v64 = rows.double().requires_grad_(True)
    seq = v64.view(B, T, C).permute(0, 2, 1)
    ref = (seq[:, :, 1:] - seq[:, :, :-1]).abs().mean()
    got = K().tv_mean_fwd(rows.to(DEV), B, C, T, T * C, 1, C)
    assert abs(got.item() - ref.item()) < 1e-6
    (gref,) = torch.autograd.grad(ref, v64)
    dx = K().tv_mean_bwd(rows.to(DEV), gout, B, C, T, T * C, 1, C)
    assert (dx.cpu().double() - 1.3 * gref).abs().max().item() < 1e-9


@pytest.mark.parametrize("L", [200, 100, 50, 25])
def test_pool_upsample(L):
    B, C = 3, 128
    x = gen(B, C, L, seed=1)
    x64 = x.double().requires_grad_(True)
    ref = F.max_pool1d(x64, 2, 2)
    got = K().maxpool2_fwd(x.to(DEV))
    assert torch.equal(got.cpu(), F.max_pool1d(x, 2, 2))
    dy = gen(*ref.shape, seed=2)
    (gref,) = torch.autograd.grad(ref, x64, dy.double())
    dx = K().maxpool2_bwd(x.to(DEV), dy.to(DEV))
    assert rel_err(dx, gref) < 1e-7
    up = torch.nn.Upsample(scale_factor=2, mode="linear", align_corners=False)
    ref = up(x64)
    got = K().upsample2_fwd(x.to(DEV))
    assert rel_err(got, ref) < 1e-6
    dy = gen(*ref.shape, seed=3)
    (gref,) = torch.autograd.grad(ref, x64, dy.double())
    dx = K().upsample2_bwd(dy.to(DEV))
    assert rel_err(dx, gref) < 1e-6
    # into a channel block of a wider buffer (the U-Net's skip concatenation made in place): same bits, neighbours untouched
    wide = torch.full((B, 2 * C, 2 * L), 7.0, device=DEV)
    K().upsample2_fwd(x.to(DEV), out=wide[:, C:])
    assert torch.equal(wide[:, C:], got) and bool((wide[:, :C] == 7.0).all())


@pytest.mark.parametrize("shape", [(2, 3, 1), (1, 2, 7), (5, 6, 3), (3, 1, 2), (4, 5, 25), (2, 7, 33)], ids=str)
def test_upsample_row_ends_in_the_flat_kernels(shape):
    """Odd lengths and rows of one to three elements: the flat-index kernels (csrc/pointwise.hip) meet a row end inside a
    thread's pair / quad at every offset; non-finite inputs propagate as in torch (0 * inf at the clamped ends)."""
    B, C, L = shape
    x = gen(B, C, L, seed=5)
    x[0, 0, 0] = float("inf")
    x[-1, -1, -1] = float("nan")
    up = torch.nn.Upsample(scale_factor=2, mode="linear", align_corners=False)
    ref = up(x)
    got = K().upsample2_fwd(x.to(DEV)).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(torch.isinf(got), torch.isinf(ref))
    assert rel_err(torch.nan_to_num(got, 0.0, 0.0, 0.0), torch.nan_to_num(ref, 0.0, 0.0, 0.0).double()) < 1e-6
    # and the flat kernels against the per-row kernels they replace where both exist (same expressions: same bits)
    if (C * L) % 2 == 0:
        rows = K().upsample2_fwd(x.view(1, B * C, L).to(DEV).view(B * C, 1, L)).view(B, C, 2 * L).cpu()   # C = 1: per-row kernel when L is odd
        assert torch.equal(torch.nan_to_num(rows, 0.0, 1e30, -1e30), torch.nan_to_num(got, 0.0, 1e30, -1e30))
    x2 = gen(B, C, L, seed=6)
    x64 = x2.double().requires_grad_(True)
    dy = gen(B, C, 2 * L, seed=7)
    (gref,) = torch.autograd.grad(up(x64), x64, dy.double())
    assert rel_err(K().upsample2_bwd(dy.to(DEV)), gref) < 1e-6


def test_profiler_counts_gemm_launches():
    k = K()
    x = gen(2, 8, 64, seed=1).to(DEV)
    w = gen(16, 8, 3, seed=2).to(DEV)
    k.prof_begin()
    k.conv1d_fwd(x, w, None, 1, 1)
    k.conv1d_fwd(x, w, None, 1, 1)
    stats = k.prof_end()
    assert stats["gemm"]["launches"] == 2
    assert stats["gemm"]["flops"] == 2 * (2.0 * 16 * (2 * 64) * (8 * 3))
    assert stats["gemm"]["ms"] > 0


@pytest.mark.parametrize("dims", [(5, 7, 250, 240, 3), (32, 9, 50, 50, 3), (3, 12, 10, 10, 1), (17, 5, 33, 21, 2)],
                         ids=lambda c: "x".join(map(str, c)))
def test_gru_stack_fwd_bwd(dims):
    """Diagonal (layer, t) schedule of a multi-layer GRU vs nn.GRU, through the autograd op."""
    from music2dance_amd import ops
    B, T, I, H, L = dims
    rnn = torch.nn.GRU(I, H, L, batch_first=True).double()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() == 1 else 1.0 / math.sqrt(p.shape[1])))
    x = gen(B, T, I, seed=1)
    x64 = x.double().requires_grad_(True)
    ref, _ = rnn(x64)
    names = [n for l in range(L) for n in ("weight_ih_l%d" % l, "weight_hh_l%d" % l, "bias_ih_l%d" % l, "bias_hh_l%d" % l)]
    params = [getattr(rnn, n).detach().float().to(DEV).requires_grad_(True) for n in names]
    xd = x.to(DEV).requires_grad_(True)
    out = ops.gru_stack(xd, params)
    assert rel_err(out, ref) < 1e-5
    dout = gen(B, T, H, seed=6)
    gref = torch.autograd.grad(ref, [x64] + [getattr(rnn, n) for n in names], dout.double())
    got = torch.autograd.grad(out, [xd] + params, dout.to(DEV))
    for a, b, n in zip(got, gref, ["x"] + names):
        assert rel_err(a, b) < 3e-5, n


def test_gru_stack_lengths():
    from music2dance_amd import ops
    B, T, I, H, L = 4, 6, 8, 16, 2
    lengths = [6, 5, 3, 1]
    rnn = torch.nn.GRU(I, H, L, batch_first=True).double()
    x = gen(B, T, I, seed=1)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x.double(), lengths, batch_first=True)
    ref, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True)
    names = [n for l in range(L) for n in ("weight_ih_l%d" % l, "weight_hh_l%d" % l, "bias_ih_l%d" % l, "bias_hh_l%d" % l)]
    params = [getattr(rnn, n).detach().float().to(DEV) for n in names]
    lens = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        out = ops.gru_stack(x.to(DEV), params, lens)
    assert rel_err(out, ref) < 1e-5


def test_conv1d_random_shapes():
    """Seeded sweep over conv geometries (tails of every kind, padding wider than the kernel
    reach, strides that do not divide, batches that split tiles across samples), plain and
    masked, against fp64 torch."""
    import random
    rng = random.Random(1234)
    k = K()
    n_done = 0
    while n_done < 48:
        B = rng.choice([1, 2, 3, 5, 9, 33])
        Cin = rng.choice([1, 2, 5, 15, 16, 17, 31, 32, 40, 64, 100])
        Cout = rng.choice([1, 3, 16, 31, 32, 33, 64, 70, 129])
        ks = rng.choice([1, 2, 3, 4, 5, 7, 9, 16, 25])
        s = rng.choice([1, 1, 2, 3, 4, 5])
        p = rng.choice([0, 0, 1, 2, ks // 2, ks - 1])
        L = rng.choice([ks, ks + 1, 8, 17, 40, 63, 130, 257])
        if L + 2 * p < ks:
            continue
        Lout = (L + 2 * p - ks) // s + 1
        seed = 100 + n_done
        x = gen(B, Cin, L, seed=seed)
        w = gen(Cout, Cin, ks, seed=seed + 1, scale=1.0 / math.sqrt(Cin * ks))
        b = gen(Cout, seed=seed + 2, scale=0.1)
        dy = gen(B, Cout, Lout, seed=seed + 3)
        mask = gen(B, Cout, Lout, seed=seed + 4)
        use_mask = n_done % 3 == 0
        tag = "B%d Cin%d L%d Cout%d k%d s%d p%d mask%d" % (B, Cin, L, Cout, ks, s, p, use_mask)
        xd, wd, bd, dyd, md = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV), mask.to(DEV)
        x64 = x.double().requires_grad_(True)
        w64 = w.double().requires_grad_(True)
        ref = F.conv1d(x64, w64, b.double(), stride=s, padding=p)
        m0 = (mask.double() > 0).double() if use_mask else torch.ones_like(ref)
        gx, gw = torch.autograd.grad(ref, (x64, w64), dy.double() * m0)
        with k.weight_cache():
            y = k.conv1d_fwd(xd, wd, bd, s, p, out_mask=md if use_mask else None, out_mask_slope=0.0)
            dx = k.conv1d_bwd_data(dyd, wd, L, s, p, dy_mask=md if use_mask else None, dy_mask_slope=0.0)
            dw = k.conv1d_bwd_weight(xd, dyd, ks, s, p, dy_mask=md if use_mask else None, dy_mask_slope=0.0)
        assert rel_err(y, ref.detach() * m0) < 2e-5, "fwd " + tag
        assert rel_err(dx, gx) < 2e-5, "bwd_data " + tag
        assert rel_err(dw, gw) < 3e-5, "bwd_weight " + tag
        n_done += 1


# ----------------------------------------------------------------------------------- window views
WIN_CASES = [
    ("default.c0", 3, 7, 640, 3200, 32, 250, 50, 124),
    ("unet.c0", 2, 5, 640, 3200, 32, 160, 4, 79),
    ("wavegan.l1-thin", 3, 6, 640, 3200, 32, 25, 4, 0),
    ("odd-geometry", 4, 9, 37, 101, 8, 5, 2, 1),
    ("one-track", 1, 12, 16, 64, 16, 7, 1, 3),
]


@pytest.mark.parametrize("case", WIN_CASES, ids=lambda c: c[0])
def test_conv1d_over_track_windows_equals_conv_on_materialised_slices(case):
    """Audio slicing fused into the first encoder conv (m2d_conv1d_fwd_windows / _bwd_weight_windows):
    same kernels, same summation order as on the materialised (B*T, 1, window) slices -> bit-equal."""
    _, B, T, hop, window, Cout, ks, s, p = case
    S = (T - 1) * hop + window
    track = gen(B, S, seed=1).to(DEV)
    w = gen(Cout, 1, ks, seed=2, scale=1.0 / math.sqrt(ks)).to(DEV)
    b = gen(Cout, seed=3, scale=0.1).to(DEV)
    slices = track.unfold(-1, window, hop)
    assert slices.shape == (B, T, window)
    dense = slices.contiguous().view(B * T, 1, window)
    k = K()
    y_w = k.conv1d_fwd_windows(track, T, hop, window, w, b, s, p, act=1)
    y_d = k.conv1d_fwd(dense, w, b, s, p, act=1)
    assert torch.equal(y_w, y_d)
    ref = F.conv1d(dense.cpu().double(), w.cpu().double(), b.cpu().double(), stride=s, padding=p).clamp_min(0)
    assert rel_err(y_w, ref) < 2e-5
    dy = gen(*y_d.shape, seed=4).to(DEV)
    if y_d.shape[2] >= 16:
        dw_w, db_w = k.conv1d_bwd_weight_windows(track, T, hop, window, dy, ks, s, p, dy_mask=y_d, dy_mask_slope=0.0,
                                                 with_bias=True)
        dw_d, db_d = k.conv1d_bwd_weight(dense, dy, ks, s, p, dy_mask=y_d, dy_mask_slope=0.0, with_bias=True)
        assert torch.equal(dw_w, dw_d) and torch.equal(db_w, db_d)
    # rows of a larger tensor (row stride > row length) are accepted as tracks
    big = gen(B, S + 13, seed=5).to(DEV)
    y_big = k.conv1d_fwd_windows(big[:, :S], T, hop, window, w, b, s, p)
    assert torch.equal(y_big, k.conv1d_fwd(big[:, :S].unfold(-1, window, hop).contiguous().view(B * T, 1, window), w, b, s, p))


# ----------------------------------------------------------------------------------- BN statistics from the conv epilogue
@pytest.mark.parametrize("case", [("enc.c1", 96, 32, 64, 64, 4, 2, 1), ("enc.c0-k250", 40, 1, 3200, 32, 250, 50, 124),
                                  ("wavegan.l1-thin", 12, 1, 3200, 32, 25, 4, 0), ("wavegan.l2", 6, 32, 794, 64, 25, 4, 0),
                                  ("unet.cb", 5, 128, 200, 128, 3, 1, 1), ("odd", 7, 19, 37, 21, 5, 2, 2),
                                  # few tiles, long K: the planner splits K; the statistics then come from the tile's last
                                  # arriver (one-launch split-K, csrc/gemm_engine.hip: fused_possible) - WaveGAN l4 at B = 32
                                  ("unet.c0-k160", 10, 1, 3200, 32, 160, 4, 79),
                                  ("wavegan.l4-splitK", 30, 128, 2560, 256, 25, 4, 11), ("splitK-8-tiles", 4, 256, 512, 256, 25, 4, 11)],
                         ids=lambda c: c[0])
def test_conv1d_epilogue_statistics_and_bn_from_sums(case):
    """conv1d_fwd(with_stats) returns the per-channel sum / sum of squares of what it stored; BatchNorm
    from those sums equals BatchNorm that reads the activation itself; backward halves equal the fused call."""
    _, B, Cin, L, Cout, ks, s, p = case
    k = K()
    x = gen(B, Cin, L, seed=1).to(DEV)
    w = gen(Cout, Cin, ks, seed=2, scale=1.0 / math.sqrt(Cin * ks)).to(DEV)
    b = gen(Cout, seed=3, scale=0.1).to(DEV)
    y, sums = k.conv1d_fwd(x, w, b, s, p, with_stats=True)
    y_plain = k.conv1d_fwd(x, w, b, s, p)
    assert sums.dtype == torch.float64 and sums.shape == (2 * Cout,)
    y64 = y.double()
    assert rel_err(sums[0::2], y64.sum((0, 2))) < 1e-6 and rel_err(sums[1::2], (y64 * y64).sum((0, 2))) < 1e-6
    assert rel_err(y, y_plain.double()) < 1e-6
    assert rel_err(k.bn_stats(y), torch.stack((y64.sum((0, 2)), (y64 * y64).sum((0, 2))), 1).reshape(-1)) < 1e-6
    g, bt = gen(Cout, seed=5).abs().add(0.5).to(DEV), gen(Cout, seed=6, scale=0.2).to(DEV)
    rm1, rv1 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    rm2, rv2 = rm1.clone(), rv1.clone()
    n = y.numel() // Cout
    o1, m1, i1 = k.bn_fwd(y, g, bt, rm1, rv1, True, 1e-5, 0.1, act=2, slope=0.2)
    o2, m2, i2 = k.bn_fwd_sums(y, sums, n, g, bt, rm2, rv2, 1e-5, 0.1, act=2, slope=0.2)
    for a_, b_ in ((o1, o2), (m1, m2), (i1, i2), (rm1, rm2), (rv1, rv2)):
        assert rel_err(a_, b_.double()) < 2e-6
    dy = gen(*y.shape, seed=7).to(DEV)
    dx1, dg1, db1 = k.bn_bwd(dy, y, g, bt, m1, i1, act=2, slope=0.2)
    loc = k.bn_bwd_stats(dy, y, g, bt, m1, i1, act=2, slope=0.2)
    dx2, dg2, db2 = k.bn_bwd_sums(dy, y, g, bt, m1, i1, loc, loc, n, act=2, slope=0.2)
    for a_, b_ in ((dx1, dx2), (dg1, dg2), (db1, db2)):
        assert rel_err(a_, b_.double()) < 2e-6
    # "two ranks": statistics over twice the data = the sums doubled, count doubled -> same normalisation
    o3, _, _ = k.bn_fwd_sums(y, 2 * sums, 2 * n, g, bt, None, None, 1e-5, 0.1, act=2, slope=0.2)
    assert rel_err(o3, o1.double()) < 2e-6


# ----------------------------------------------------------------------------------- persistent GRU
@pytest.mark.parametrize("dims", [(64, 120, 240, 3), (16, 300, 240, 3), (32, 120, 50, 3), (64, 120, 10, 1), (5, 7, 33, 2)],
                         ids=lambda d: "B%d-T%d-H%d-L%d" % d)
def test_persistent_gru_is_bit_identical_to_the_step_launches(dims):
    """One persistent launch (weights in LDS, cross-CU hand-off per step through write-through stores and
    agent-scope counters) against T + L - 1 dependent step launches: same MFMA and reduction order, so every
    word must be equal - outputs and the gates saved for BPTT. Run twice more under a concurrent stream of
    large GEMMs (uneven load, other CUs' L1 / L2 busy) to stress the hand-off."""
    B, T, H, L = dims
    k = K()
    g = torch.Generator().manual_seed(3)
    gi0 = (torch.randn(B, T, 3 * H, generator=g) * 0.5).to(DEV)
    w_hh_t = [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L)]
    w_ih_t = [None] + [(torch.randn(H, 3 * H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L - 1)]
    b_hh = [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L)]
    b_ih = [None] + [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L - 1)]
    lens = None if B != 5 else torch.tensor([7, 6, 4, 2, 1], dtype=torch.int32, device=DEV)
    ref_o, ref_s = k.gru_stack_fwd(gi0, w_ih_t, b_ih, w_hh_t, b_hh, lens, True, persistent=False)
    a = torch.randn(4096, 4096, device=DEV)
    side = torch.cuda.Stream()
    for trial in range(3):
        if trial:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    k.gemm(0, a, a)
        out, saved = k.gru_stack_fwd(gi0, w_ih_t, b_ih, w_hh_t, b_hh, lens, True, persistent=True)
        torch.cuda.synchronize()
        k.check_async_errors()
        for l in range(L):
            assert torch.equal(out[l], ref_o[l]), (trial, l)
            assert torch.equal(saved[l], ref_s[l]), (trial, l)


@pytest.mark.parametrize("dims", [(64, 120, 240, 3), (16, 300, 240, 3), (32, 120, 50, 3), (64, 120, 10, 1), (5, 7, 33, 2)],
                         ids=lambda d: "B%d-T%d-H%d-L%d" % d)
def test_persistent_gru_backward_is_bit_identical_to_the_step_launches(dims):
    """Back-propagation through time as ONE persistent launch (round 3) against the T + L - 1 anti-diagonal step
    launches: same k-step assignment, MFMA order and cross-wave sum, so dgi / dgh of every layer must be equal word
    for word - alone and under a concurrent stream of large GEMMs."""
    B, T, H, L = dims
    k = K()
    g = torch.Generator().manual_seed(4)
    gi0 = (torch.randn(B, T, 3 * H, generator=g) * 0.5).to(DEV)
    w_hh = [(torch.randn(3 * H, H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L)]
    w_ih = [None] + [(torch.randn(3 * H, H, generator=g) / math.sqrt(H)).to(DEV) for _ in range(L - 1)]
    b_hh = [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L)]
    b_ih = [None] + [(torch.randn(3 * H, generator=g) * 0.1).to(DEV) for _ in range(L - 1)]
    lens = None if B != 5 else torch.tensor([7, 6, 4, 2, 1], dtype=torch.int32, device=DEV)
    outs, saved = k.gru_stack_fwd(gi0, [None] + [w.t().contiguous() for w in w_ih[1:]], b_ih,
                                  [w.t().contiguous() for w in w_hh], b_hh, lens, True, persistent=False)
    dout = torch.randn(B, T, H, generator=g).to(DEV)
    ref_gi, ref_gh = k.gru_stack_bwd(dout, outs, saved, w_hh, w_ih, lens, persistent=False)
    a = torch.randn(4096, 4096, device=DEV)
    side = torch.cuda.Stream()
    for trial in range(3):
        if trial:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    k.gemm(0, a, a)
        dgi, dgh = k.gru_stack_bwd(dout, outs, saved, w_hh, w_ih, lens, persistent=True)
        torch.cuda.synchronize()
        k.check_async_errors()
        for l in range(L):
            assert torch.equal(dgi[l], ref_gi[l]), (trial, l)
            assert torch.equal(dgh[l], ref_gh[l]), (trial, l)


# ------------------------------------------------------------------------------ round 3: critic-step entry points
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gemm_on_column_blocks_of_wider_buffers(mode):
    """m2d_gemm_ld: operands / output / masks are column blocks (row pitch > row length) of wider buffers."""
    M, N, Kd = 37, 100, 128
    g = torch.Generator().manual_seed(mode)
    wide = lambda r, c, pad: torch.randn(r, c + pad, generator=g).to(DEV)
    if mode == 0:
        A, Bm = wide(M, Kd, 5), wide(N, Kd, 9)
        a, b = A[:, 2:2 + Kd], Bm[:, 4:4 + Kd]
        ref = a.double().cpu() @ b.double().cpu().t()
    elif mode == 1:
        A, Bm = wide(M, Kd, 5), wide(Kd, N, 100)
        a, b = A[:, 2:2 + Kd], Bm[:, 100:]
        ref = a.double().cpu() @ b.double().cpu()
    else:
        A, Bm = wide(Kd, M, 3), wide(Kd, N, 100)
        a, b = A[:, 1:1 + M], Bm[:, :N]
        ref = a.double().cpu().t() @ b.double().cpu()
    C = torch.full((M, N + 60), 7.0, device=DEV)
    out = C[:, 20:20 + N]
    got = K().gemm_ld(mode, a, b, out=out)
    assert got.data_ptr() == out.data_ptr()
    assert rel_err(out, ref) < 2e-5
    assert float((C[:, :20] - 7).abs().max()) == 0 and float((C[:, 20 + N:] - 7).abs().max()) == 0
    # operand mask sharing a's pitch, output mask in place (the output block holds the mask values on entry)
    if mode != 2:
        am = torch.randn(A.shape, generator=g).to(DEV)
        amv = am[:, 2:2 + Kd]
        C2 = torch.randn(M, N + 60, generator=g).to(DEV)
        om = C2[:, 20:20 + N].clone()
        K().gemm_ld(mode, a, b, a_mask=amv, a_mask_slope=0.0, out_mask=C2[:, 20:20 + N], out_mask_slope=0.0,
                    out=C2[:, 20:20 + N])
        am64 = a.double().cpu() * (amv.double().cpu() > 0)
        ref2 = (am64 @ (b.double().cpu().t() if mode == 0 else b.double().cpu())) * (om.double().cpu() > 0)
        assert rel_err(C2[:, 20:20 + N], ref2) < 2e-5


@pytest.mark.parametrize("dims", [(5, 120, 69), (3, 33, 69), (2, 300, 69), (4, 7, 5)], ids=lambda d: "x".join(map(str, d)))
def test_pose_pack3_is_bit_exact(dims):
    B, T, C = dims
    g = torch.Generator().manual_seed(B)
    real, fake, alpha = torch.rand(B, T, C, generator=g), torch.randn(B * T, C, generator=g), torch.rand(B, generator=g)
    out = K().pose_pack3(real.to(DEV), fake.to(DEV), alpha.to(DEV)).cpu()
    a = alpha.view(B, 1, 1)
    interp = a * real + (1 - a) * fake.view(B, T, C)  # the reference's expression (losses.py:20), fp32
    want = torch.cat((interp, real, fake.view(B, T, C)), 0).permute(0, 2, 1).contiguous()
    assert torch.equal(out, want)


def test_wgan_critic_loss_scalars():
    B = 37
    g = torch.Generator().manual_seed(1)
    s = torch.randn(3 * B, generator=g)
    p0, p1 = torch.rand((), generator=g), torch.rand((), generator=g)
    out = K().wgan_critic_loss(s.to(DEV), B, p0.to(DEV), p1.to(DEV), 10.0).cpu().double()
    w = s[2 * B:].double().mean() - s[B:2 * B].double().mean()
    gp = p0.double() + p1.double()
    assert abs(out[2] - w) < 1e-6 and abs(out[1] - gp) < 1e-6 and abs(out[0] - (w + 10 * gp)) < 1e-5
    out = K().wgan_critic_loss(s.to(DEV), B, p0.to(DEV), None, 10.0).cpu().double()
    assert abs(out[1] - p0.double()) < 1e-7


# ----------------------------------------------------------------------------------- split-K in one launch
def test_split_k_in_one_launch_equals_the_two_launch_form():
    """Round 3: with the stream's ticket scratch registered (include/m2d.h: m2d_stream_scratch_set; kernels.py does it
    on a stream's first launch) a split-K plan finishes inside the GEMM launch; without it the C-ABI falls back to
    GEMM + m2d_splitk_reduce_kernel. Same operands, both forms, against fp64: a weight gradient (K = B * Lout = 15 360,
    a few tiles: always split) and a small-N forward."""
    from music2dance_amd import _lib, kernels
    k = K()
    B, Cin, L, Cout, ks, s, p = 128, 128, 120, 128, 7, 1, 3
    x = gen(B, Cin, L, seed=1).to(DEV)
    dy = gen(B, Cout, L, seed=2).to(DEV)
    w = (gen(Cout, Cin, ks, seed=3) / math.sqrt(Cin * ks)).to(DEV)
    bias = gen(Cout, seed=4).to(DEV)
    ref_w = torch.nn.grad.conv1d_weight(x.double().cpu(), w.shape, dy.double().cpu(), stride=s, padding=p)
    ref_y = F.conv1d(x[:8].double().cpu(), w.double().cpu(), bias.double().cpu(), stride=s, padding=p)
    stream = kernels._stream(torch.device(DEV))  # registers the scratch
    key = (torch.device(DEV).index, stream)
    assert key in kernels._STREAM_SCRATCH
    k.prof_begin()
    gw_fused = k.conv1d_bwd_weight(x, dy, ks, s, p)
    y_fused = k.conv1d_fwd(x[:8].contiguous(), w, bias, s, p)
    torch.cuda.synchronize()
    k.prof_end()
    # two-launch form: unregister, run, register again
    t = kernels._STREAM_SCRATCH[key]
    _lib.check(_lib.lib().m2d_stream_scratch_set(stream, 0, 0), "m2d_stream_scratch_set")
    try:
        gw_two = k.conv1d_bwd_weight(x, dy, ks, s, p)
        y_two = k.conv1d_fwd(x[:8].contiguous(), w, bias, s, p)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().m2d_stream_scratch_set(stream, t.data_ptr(), t.numel() * 4), "m2d_stream_scratch_set")
    assert int(t.abs().sum().item()) == 0  # the tickets are left zero
    assert rel_err(gw_fused, ref_w) <= 2e-5 and rel_err(gw_two, ref_w) <= 2e-5
    assert rel_err(y_fused, ref_y) <= 2e-5 and rel_err(y_two, ref_y) <= 2e-5
    assert rel_err(gw_fused, gw_two.double()) <= 2e-6
    # and again on the registered stream: the tickets a launch left behind serve the next one
    gw_again = k.conv1d_bwd_weight(x, dy, ks, s, p)
    assert torch.equal(gw_again, gw_fused)


def test_a_timed_out_recurrence_voids_the_queued_optimizer_steps_and_the_run_goes_on():
    """Recovery without leaving the process (round-3 verdict): the word a persistent GRU launch raises when it gives up is
    the `skip` flag of both optimizers - an Adam step queued behind it is a no-op on the device; the engine then finds the
    word, synchronises, clears it, counts the fault and runs the recurrences as per-step launches from there on."""
    import bench
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
    dev = torch.device("cuda:0")
    k = K()
    gen, critic = bench.build_models(dev, 120)
    eng = Phase3Engine(gen, critic, dict(bench.P3_DEFAULT, n_critic_steps=2), data_parallel=False)
    real, audio, slices = synthetic_phase3_batch(4, 120, dev, seed=3)
    try:
        assert eng.optim_critic.skip_flag is not None and type(k).persistent_gru
        for _ in range(2):
            eng.train_step(real, audio, slices)
        torch.cuda.synchronize()
        before = [p.detach().clone() for p in list(critic.parameters())[:4]]
        faults = type(k).async_faults
        # gradients are in place from the last iteration; raise the word as a persistent launch that gave up would (on
        # the device it is raised by a kernel that ran BEFORE the optimizer step queued behind it - from the host the
        # closest stand-in is: raise, then queue the step)
        for p in critic.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        k.raise_async_fault()
        eng.optim_critic.step()
        torch.cuda.synchronize()
        after = [p.detach() for p in list(critic.parameters())[:4]]
        assert all(torch.equal(a, b) for a, b in zip(before, after)), "an optimizer step ran although the fault word was raised"
        with pytest.warns(UserWarning, match="timed out"):
            eng.train_step(real, audio, slices)     # the engine finds the word at its next check: recovers, goes on
        assert type(k).async_faults == faults + 1 and not type(k).persistent_gru
        for _ in range(2):
            out = eng.train_step(real, audio, slices)   # step launches now
        eng.flush()
        assert all(torch.isfinite(v).all() for v in out.values())
        assert not all(torch.equal(a, b.detach()) for a, b in zip(before, list(critic.parameters())[:4]))
    finally:
        type(k).persistent_gru = True


def test_a_recovery_drops_the_captured_graphs_and_they_are_recaptured_without_the_persistent_kernel():
    """ADVICE (round 4): the fault recovery flips `persistent_gru`, which only the eager launches consult - graphs
    captured earlier have the persistent recurrent kernel baked in and would replay (and time out) for ever. After a
    recovery the engine drops them; the next train_step re-captures with per-step recurrent launches."""
    import bench
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
    dev = torch.device("cuda:0")
    k = K()
    gen, critic = bench.build_models(dev, 120)
    eng = Phase3Engine(gen, critic, dict(bench.P3_DEFAULT, n_critic_steps=2), data_parallel=False).enable_graphs()
    real, audio, slices = synthetic_phase3_batch(4, 120, dev, seed=3)
    try:
        assert type(k).persistent_gru
        for _ in range(2):
            eng.train_step(real, audio, slices)
        torch.cuda.synchronize()
        assert len(eng._graphs) == 1
        old = next(iter(eng._graphs.values()))
        k.raise_async_fault()
        with pytest.warns(UserWarning, match="timed out"):
            eng.train_step(real, audio, slices)
        assert not type(k).persistent_gru
        assert len(eng._graphs) == 1 and next(iter(eng._graphs.values())) is not old, "the stale graph was replayed"
        for _ in range(2):
            out = eng.train_step(real, audio, slices)
        eng.flush()
        assert all(torch.isfinite(v).all() for v in out.values())
    finally:
        type(k).persistent_gru = True


@pytest.mark.parametrize("cin,cout,ks,s,p,L", [(32, 64, 25, 4, 11, 384), (64, 128, 25, 4, 11, 256), (256, 512, 25, 4, 11, 300),
                                                 (128, 128, 7, 1, 3, 120), (69, 128, 25, 1, 12, 120)])
@pytest.mark.parametrize("B,mb", [(6, 3), (5, 3), (128, 64)])
def test_backward_data_with_a_shared_mask_equals_the_repeated_mask(cin, cout, ks, s, p, L, B, mb):
    """m2d_conv1d_bwd_data_shared_mask (round 5): the mask holds the first `mb` samples and the samples behind them read
    it from its start again - bit-equal to the ordinary call with the mask tensor repeated (sub-pixel, polyphase,
    stride-1 / 16-byte and split-K epilogues)."""
    if B == 128 and L > 256:
        pytest.skip("large case on the short layers only")
    k = K()
    g = torch.Generator().manual_seed(5)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(B, cout, Lout, generator=g).to(DEV)
    w = (torch.randn(cout, cin, ks, generator=g) / math.sqrt(cin * ks)).to(DEV)
    # (the mask is the FRONT of a larger buffer whose tail is poison: a read past the shared mask shows)
    buf = torch.full((mb + 1, cin, L), float("nan"), device=DEV)
    buf[:mb] = torch.randn(mb, cin, L, generator=g).to(DEV)
    mask = buf[:mb]
    full = torch.cat([mask, mask[:B - mb]], 0).contiguous()
    for slope in (0.0, 0.2):
        with k.weight_cache():
            want = k.conv1d_bwd_data(dy, w, L, s, p, out_mask=full, out_mask_slope=slope)
            got = k.conv1d_bwd_data(dy, w, L, s, p, out_mask=mask, out_mask_slope=slope)
        assert torch.equal(got, want), (got - want).abs().max()


@pytest.mark.parametrize("cin,cout,L", [(32, 64, 384), (32, 64, 19200), (64, 128, 4800), (64, 48, 260)])
@pytest.mark.parametrize("B", [3, 64])
def test_phase_major_sub_pixel_backward_data(cin, cout, L, B):
    """m2d_gemm_dl_tall_kernel (round 5): the k25 / stride-4 backward-data of a 32- or 64-channel layer in its phase-major
    sub-pixel form, the last tap slot multiplied for phase 0 only - against fp64 autograd, against the form it replaces (a
    dy mask of ones sends the same problem through the register-staging kernel with the (ci, r) / polyphase layout), with
    an output mask, a residual and ragged row ends (L = 260: the last quad of a row hangs over its end)."""
    if B == 64 and L > 4800:
        pytest.skip("the full-length case at B = 3 only")
    k = K()
    ks, s, p = 25, 4, 11
    g = torch.Generator().manual_seed(11)
    Lout = (L + 2 * p - ks) // s + 1
    dy = torch.randn(B, cout, Lout, generator=g).to(DEV)
    w = (torch.randn(cout, cin, ks, generator=g) / math.sqrt(cin * ks)).to(DEV)
    mask = torch.randn(B, cin, L, generator=g).to(DEV)
    res = torch.randn(B, cin, L, generator=g).to(DEV)
    ref = torch.nn.grad.conv1d_input((B, cin, L), w.double(), dy.double(), stride=s, padding=p)
    with k.weight_cache():
        got = k.conv1d_bwd_data(dy, w, L, s, p)
        old = k.conv1d_bwd_data(dy, w, L, s, p, dy_mask=torch.ones_like(dy), dy_mask_slope=0.0)
        got_m = k.conv1d_bwd_data(dy, w, L, s, p, out_mask=mask, out_mask_slope=0.2)
        got_r = k.conv1d_bwd_data(dy, w, L, s, p, out_mask=mask, out_mask_slope=0.0, residual=res)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 2e-6 * scale
    assert (got - old).abs().max().item() < 2e-6 * scale
    m02 = torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, 0.2)).double()
    assert (got_m.double() - ref * m02).abs().max().item() < 2e-6 * scale
    want_r = (ref + res.double()) * (mask > 0).double()
    assert (got_r.double() - want_r).abs().max().item() < 2e-6 * max(scale, want_r.abs().max().item())


@pytest.mark.parametrize("shape", [(9, 16, 200), (5, 8, 50), (4, 6, 25), (7, 6, 3), (3, 5, 10), (40, 128, 100)], ids=str)
def test_batchnorm_pass_with_the_following_pool_or_upsampling_fused_in(shape):
    """kernels.bn_fwd_sums_pool / bn_fwd_sums_upsample2 (the U-Net without an autograd graph, csrc/bn.hip round 6): the same
    bits as bn_fwd_sums followed by the separate max-pool / upsampling kernels - outputs, save_mean / save_invstd and the
    running buffers (the finalisation of the statistics runs inside the normalisation launch) - into dense outputs and into
    channel blocks of a wider buffer."""
    B, C, L = shape
    k = K()
    x = gen(B, C, L, seed=1).to(DEV)
    g, b = gen(C, seed=2).abs().add(0.5).to(DEV), gen(C, seed=3, scale=0.3).to(DEV)
    x64 = x.double()
    sums = torch.stack((x64.sum((0, 2)), (x64 * x64).sum((0, 2))), 1).reshape(-1)

    def buffers():
        return torch.full((C,), 0.25, device=DEV), torch.full((C,), 1.5, device=DEV)
    rm1, rv1 = buffers()
    y_ref, m_ref, i_ref = k.bn_fwd_sums(x, sums, float(B * L), g, b, rm1, rv1, 1e-5, 0.1, act=2, slope=0.2)
    if (C * L) % 2 == 0:
        up_ref = k.upsample2_fwd(y_ref)
        rm2, rv2 = buffers()
        up, m2, i2 = k.bn_fwd_sums_upsample2(x, sums, float(B * L), g, b, rm2, rv2, 1e-5, 0.1, act=2, slope=0.2)
        assert torch.equal(up, up_ref) and torch.equal(m2, m_ref) and torch.equal(i2, i_ref)
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
        if (2 * C * 2 * L) % 4 == 0:
            wide = torch.full((B, 2 * C, 2 * L), 7.0, device=DEV)
            k.bn_fwd_sums_upsample2(x, sums, float(B * L), g, b, *buffers(), 1e-5, 0.1, act=2, slope=0.2, out=wide[:, :C])
            assert torch.equal(wide[:, :C], up_ref) and bool((wide[:, C:] == 7.0).all())
    if L % 2 == 0:
        wide = torch.full((B, 2 * C, L), 7.0, device=DEV)
        rm3, rv3 = buffers()
        y, p, m3, i3 = k.bn_fwd_sums_pool(x, sums, float(B * L), g, b, rm3, rv3, 1e-5, 0.1, act=2, slope=0.2, out=wide[:, C:])
        assert torch.equal(y, y_ref) and torch.equal(p, k.maxpool2_fwd(y_ref)) and bool((wide[:, :C] == 7.0).all())
        assert torch.equal(m3, m_ref) and torch.equal(i3, i_ref) and torch.equal(rm1, rm3) and torch.equal(rv1, rv3)
