"""The flip audit (tests/flip_audit.py) at fixture size, on both kernel layers: for the generator iteration of phase 3
(generator forward, frozen critic on the generated poses), every activation mask the product uses that differs from the
fp32 oracle's is a verified rounding flip, and against the oracle evaluated WITH THE PRODUCT'S MASKS every generator
gradient element agrees under a strict bound - no flip slack, no floor for tensors that carry gradient.
The same machinery runs at the BASELINE sizes in tests/test_gpu_full_size_parity.py."""
import pytest
import torch

from music2dance_amd import kernels, ops
from music2dance_amd.losses import tv_loss
from music2dance_amd.phase3.archis.default import AblatedSequenceDiscriminator, SequenceDiscriminator, SequenceGenerator
from oracle import m2d_oracle as O
from tests.flip_audit import audit, oracle_trace, product_masks
from tests.golden import patterns as P


@pytest.fixture(params=["cpu-fake", pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    if request.param == "hip":
        assert kernels.impl().name == "hip"
        yield torch.device("cuda:0")
        return
    from tests.fake_backend import FakeKernels
    prev = kernels.set_impl(FakeKernels())
    yield torch.device("cpu")
    kernels.set_impl(prev)


def generator_iteration_oracle(sd0, d_params, slices, noise, real_c, audio_c, enc, ablated, dtype, impose=None, grad=False,
                               beta=1.0, eta=0.5):
    """The oracle's generator iteration (phase3/train.py:222-237) from BatchNorm state `sd0` under an oracle_trace.
    -> (trace, loss, {generator parameter: gradient} or None)"""
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t.clone()
    sd = {k: cast(v.detach()) for k, v in sd0.items()}
    g_params = {k: v.requires_grad_(grad) for k, v in sd.items() if O.is_param(k)}
    dp = {k: cast(v.detach()) for k, v in d_params.items()}
    B, T = real_c.shape[0], real_c.shape[2]
    cfg = O.P3Config(enc_type=enc, ablated=ablated)
    crit = lambda x: O.p3_critic(dp, x, None if ablated else cast(audio_c), cfg.init_ker, cfg.activ, ablated)
    with oracle_trace(impose=impose) as tr, torch.set_grad_enabled(grad):
        rows = O.p3_generator(sd, cast(slices), cast(noise), enc, "id", 3, 2, True)
        fake = rows.view(B, T, 69).permute(0, 2, 1)
        rc = cast(real_c)
        l1 = (rc - fake).abs().mean()
        loss = crit(rc).mean() - crit(fake).mean() + beta * l1 + eta * O.tv_loss(fake)
        grads = O.grads_of(loss, g_params) if grad else None
    return tr, loss.detach(), grads


@pytest.mark.parametrize("enc,ablated", [("default", False), ("wavegan", False), ("wavegan", True)])
def test_generator_gradients_against_the_oracle_with_the_products_masks(dev, enc, ablated):
    B, T = 2, 120
    torch.manual_seed(0)
    gen = SequenceGenerator(P.WINDOW, 250, 250, 256, 69, 10, 2, 3, enc, "id", "cpu")
    cls = AblatedSequenceDiscriminator if ablated else SequenceDiscriminator
    critic = cls(69, 128, 100, T, init_ker=25, activ="id", device="cpu")
    sd0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    d_params, _ = O.split_state({k: v.detach().clone() for k, v in critic.state_dict().items()})
    real, aud, nz = P.poses(B, T, seed=41), P.audio(B, T, seed=42), P.noise(B, T, 10, seed=43)
    sl = P.slices(aud)
    real_c, audio_c = real.permute(0, 2, 1).contiguous(), aud.unsqueeze(1)

    # ---- product: generator forward + frozen critic on real / fake, masks recorded
    gen.to(dev).train(), critic.to(dev)
    with product_masks(gen, critic) as pm:
        rows = gen(sl.to(dev), [T] * B, nz.to(dev))
        fake = rows.view(B, T, 69).permute(0, 2, 1)
        rc = real_c.to(dev)
        D = (lambda x: critic(x)) if ablated else (lambda x: critic(x, audio_c.to(dev)))
        with torch.no_grad():
            e_real = D(rc).mean()
        l1 = ops.l1_mean(real.reshape(B * T, 69).to(dev), rows)
        loss = e_real - D(fake).mean() + 1.0 * l1 + 0.5 * tv_loss(fake)
        for p in critic.parameters():
            p.requires_grad_(False)
        loss.backward()

    # ---- oracle: fp32 and fp64 traces of the same forwards, then the fp32 run with the product's masks imposed
    t32, l32, _ = generator_iteration_oracle(sd0, d_params, sl, nz, real_c, audio_c, enc, ablated, torch.float32)
    t64, _, _ = generator_iteration_oracle(sd0, d_params, sl, nz, real_c, audio_c, enc, ablated, torch.float64)
    assert set(t32.sites) == set(t64.sites) and set(t32.sites) <= set(pm.masks), sorted(set(t32.sites) - set(pm.masks))
    flips, worst, per = audit(pm.masks, t32.sites, t64.sites)
    _, l_imp, g_imp = generator_iteration_oracle(sd0, d_params, sl, nz, real_c, audio_c, enc, ablated, torch.float32,
                                                  impose=pm.masks, grad=True)
    assert abs(float(loss.detach()) - float(l_imp)) <= 1e-4 + 1e-5 * abs(float(l_imp))
    # strict: every element of every tensor that carries gradient within 2e-4 of the tensor's largest element
    emax = max(g.abs().max().item() for g in g_imp.values() if g is not None)
    worst_e = 0.0
    for name, p in gen.named_parameters():
        rg = g_imp.get(name)
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        diff = (p.grad.detach().cpu().double() - rg.double()).abs().max().item()
        scale = rg.abs().max().item()
        if scale > 1e-3 * emax:
            worst_e = max(worst_e, diff / scale)
            assert diff <= 2e-4 * scale, "%s: %.3e of the tensor's largest element (flips: %d)" % (name, diff / scale, flips)
        else:   # zero in exact arithmetic (a conv bias in front of BatchNorm): rounding noise on both sides
            assert diff <= 1e-4 * emax + 2e-3 * scale, name
    print("%s %s: %d mask flips (worst %.1e of the layer's scale from the kink), worst gradient element %.2e" %
          (enc, dev.type, flips, worst, worst_e))
