"""Pins the CPU oracle (oracle/m2d_oracle.py) to the reference: every function of the
restatement is compared with outputs of clementabary/music2dance itself, captured by
tests/golden/make_golden.py in the build container (tests/golden/*.npz).

Tolerances: 2e-5 abs on poses / scores / penalties (the reference's own fp32 noise floor
is ~1e-5), 2e-3 relative on per-parameter gradient norms and
post-Adam checksums. Loss traces over 8 optimiser steps: the synthetic critic is far from
1-Lipschitz, so gamma*GP drives losses to O(100) within a few steps and Adam turns fp32
rounding noise into O(lr) parameter moves; the reference-vs-oracle gap grows from 1e-6 at
step 1 to ~1e-2 at step 8. The traces therefore pin
the loop semantics (gating, loss formulas, Adam) at 2e-2 relative, and exactly at
step 1.
"""
import os

import numpy as np
import pytest
import torch

from oracle import m2d_oracle as O
from tests.golden import patterns as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def filled(fx, which, seed):
    sd = P.fill_state_dict(P.template(fx[which + "_keys"], fx[which + "_shapes"]), seed)
    np.testing.assert_allclose(P.sd_checksums(sd), fx[which + "_sd_sum"], rtol=1e-12, atol=1e-12,
                               err_msg="seeded weight pattern drifted (torch RNG?)")
    return sd


def close(got, want, atol=2e-5, rtol=0.0):
    got = got.detach().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, want, atol=atol, rtol=rtol)


def norms_close(grads, params_order, want, rtol=2e-3):
    got = np.array([float("nan") if grads.get(k) is None else grads[k].double().norm().item() for k in params_order])
    assert np.array_equal(np.isnan(got), np.isnan(want)), "set of parameters without gradient differs"
    m = ~np.isnan(want)
    # parameters whose true gradient is zero (a bias in front of a BatchNorm) carry pure
    # rounding noise: compare those on the scale of the largest gradient
    np.testing.assert_allclose(got[m], want[m], rtol=rtol, atol=1e-6 * np.nanmax(want))


def sums_close(sd, want, rtol=2e-4, adam_lr=0.0, adam_steps=0):
    """Checksums (sum, sum|.|) per tensor. After Adam steps a parameter whose gradient is
    pure rounding noise (a bias in front of a BatchNorm) still moves by ~lr per step in a
    noise-determined direction, so the absolute tolerance scales with lr * steps * sqrt(n)."""
    got = P.sd_checksums(sd)
    numel = np.array([v.numel() for v in sd.values()], dtype=np.float64)
    atol = 1e-5 + 4.0 * adam_lr * adam_steps * np.sqrt(numel)
    err = np.abs(got - want)
    bound = atol[:, None] + rtol * np.abs(want)
    assert (err <= bound).all(), "checksum mismatch at tensors %s" % [
        (list(sd)[i], got[i].tolist(), want[i].tolist()) for i in np.nonzero((err > bound).any(1))[0][:5]]


def param_names(sd):
    return [k for k in sd if O.is_param(k)]


# ------------------------------------------------------------------------------ phase 1
def test_p1_forward_and_critic_iteration():
    fx = load("p1")
    B = 8
    gsd, dsd = filled(fx, "gen", 1000), filled(fx, "critic", 2000)
    z = P.noise(B, 1, 10, seed=21).view(B, 10)
    real = P.poses(B, 1, seed=22).view(B, 23, 3)
    close(np.array(P.checksum(z)), fx["z_sum"], 1e-9)
    close(O.p1_generator(dict(gsd), z, 1, False), fx["gen_eval"])
    close(O.p1_critic(dsd, real, 1), fx["critic_eval"])
    v = O.p1_critic_iteration_values(gsd, dsd, z, real, 5)
    close(v["fake"], fx["gen_train"])
    assert abs(v["gp"] - fx["gp"]) < 2e-5 and abs(v["err_real"] - fx["err_real"]) < 2e-5
    assert abs(v["err_fake"] - fx["err_fake"]) < 2e-5
    norms_close(v["grads"], param_names(dsd), fx["critic_grad_norms"])
    sums_close({k: v["gen_buffers"][k] for k in v["gen_buffers"] if "running" in k}, fx["gen_bn_after"])


def test_p1_trace():
    fx = load("p1")
    gsd, dsd = filled(fx, "gen", 1000), filled(fx, "critic", 2000)
    real = P.poses(8, 1, seed=22).view(8, 23, 3)
    tr, g_out, d_out = O.p1_train_iterations(gsd, dsd, real, 6, 6)
    close(np.array(tr["loss_critic"]), fx["trace_loss_critic"], 1e-4)
    close(np.array(tr["loss_gen"]), fx["trace_loss_gen"], 1e-4)
    sums_close({k: g_out[k] for k in gsd}, fx["gen_final_sum"], adam_lr=1e-4, adam_steps=1)
    sums_close({k: d_out[k] for k in dsd}, fx["critic_final_sum"], adam_lr=1e-4, adam_steps=6)


# ------------------------------------------------------------------------------ phase 2
def test_p2_forward_lp_grads():
    fx = load("p2")
    B, T = 2, 120
    gsd, dsd = filled(fx, "gen", 3000), filled(fx, "critic", 4000)
    noise, real = P.noise(B, T, 50, seed=31), P.poses(B, T, seed=32)
    real_c = real.permute(0, 2, 1).contiguous()
    g_params, g_buf = O.split_state(gsd)
    sd = dict(g_params)
    sd.update(g_buf)
    rows = O.p2_generator(sd, noise, 3, 2, True)
    close(rows, fx["gen_train"])
    sums_close({k: sd[k] for k in gsd if "running" in k}, fx["gen_bn_after"])
    # the fixture's eval-mode forwards ran after the train-mode one (advanced running stats)
    with torch.no_grad():
        close(O.p2_generator(dict(sd), noise, 3, 2, False), fx["gen_eval"])
        close(O.p2_generator(dict(sd), noise, 3, 2, False, lengths=[T, 100]), fx["gen_eval_lengths"])
    fake = rows.view(B, T, 69).permute(0, 2, 1).contiguous()
    d_params, _ = O.split_state(dsd)
    D = lambda x: O.p2_critic(d_params, x, 3, 25)
    close(D(real_c), fx["score_real"])
    close(D(fake.detach()), fx["score_fake"])
    torch.manual_seed(7)
    alpha = torch.rand(B, 1)
    close(alpha, fx["alpha"], 0)
    lp, _, _ = O.gradient_penalty(D, real_c, fake, alpha, None, True, True)
    gp, _, _ = O.gradient_penalty(D, real_c, fake, alpha, None, True, False)
    assert abs(lp.item() - fx["lp"]) < 2e-5 and abs(gp.item() - fx["gp"]) < 2e-5
    err_critic = D(fake.detach()).mean() - D(real_c).mean() + 10 * lp
    norms_close(O.grads_of(err_critic, d_params), param_names(dsd), fx["critic_grad_norms"])
    fake_g = rows.view(B, T, 69).permute(0, 2, 1)
    err_gen = D(real_c).mean() - D(fake_g).mean() + 50 * O.tv_loss(fake_g)
    assert abs(err_gen.item() - fx["err_gen"]) < 1e-4 and abs(O.tv_loss(fake_g).item() - fx["tv"]) < 1e-6
    norms_close(O.grads_of(err_gen, g_params), param_names(gsd), fx["gen_grad_norms"])


def test_p2_trace():
    fx = load("p2")
    gsd, dsd = filled(fx, "gen", 3000), filled(fx, "critic", 4000)
    real = P.poses(2, 120, seed=32)
    lr = float(fx["trace_lr"])  # 5e-5: see make_golden.py::case_p2 (at the config's 5e-4 the trace diverges)
    tr, g_out, d_out = O.p2_train_iterations(gsd, dsd, real, 8, 8, lr=lr)
    for k in ("loss_critic", "gp", "w_dist", "loss_gen"):
        close(np.array(tr[k]), fx["trace_" + k], 1e-3, 2e-3)
    close(np.array(tr["loss_critic"][:1]), fx["trace_loss_critic"][:1], 2e-5)
    assert tr["g_step"] == [0] * 7 + [1]
    sums_close({k: g_out[k] for k in gsd}, fx["gen_final_sum"], adam_lr=lr, adam_steps=1)
    sums_close({k: d_out[k] for k in dsd}, fx["critic_final_sum"], adam_lr=lr, adam_steps=8)


# ------------------------------------------------------------------------------ phase 3
P3_CASES = [("default", "id", False, 120, 2), ("default", "tanh", False, 120, 2), ("default", "relu", True, 120, 2),
            ("wavegan", "id", False, 120, 2), ("wavegan", "tanh", True, 120, 2), ("unet", "id", False, 120, 2),
            ("unet", "id", True, 120, 2), ("unet", "id", True, 300, 1)]


def p3_name(enc, activ, ablated, T):
    return "p3_%s_%s_%s%s" % (enc, activ, "abl" if ablated else "full", "" if T == 120 else "_T%d" % T)


@pytest.mark.parametrize("case", P3_CASES, ids=lambda c: p3_name(*c[:4]))
def test_p3_forward_gp_grads(case):
    enc, activ, ablated, T, B = case
    fx = load(p3_name(enc, activ, ablated, T))
    gsd, dsd = filled(fx, "gen", 5000), filled(fx, "critic", 6000)
    real, aud, nz = P.poses(B, T, seed=41), P.audio(B, T, seed=42), P.noise(B, T, 10, seed=43)
    close(np.array(P.checksum(aud)), fx["audio_sum"], 1e-9)
    sl = O.slice_audio(aud, P.WINDOW, P.HOP, P.PAD)
    assert torch.equal(sl, P.slices(aud))
    real_c = real.permute(0, 2, 1).contiguous()
    audio_c = aud.unsqueeze(1)
    g_params, g_buf = O.split_state(gsd)
    sd = dict(g_params)
    sd.update(g_buf)
    rows = O.p3_generator(sd, sl, nz, enc, activ, 3, 2, True)
    close(rows, fx["gen_train"])
    with torch.no_grad():  # eval-mode forward after the train-mode one, like the fixture
        close(O.p3_generator(dict(sd), sl, nz, enc, activ, 3, 2, False), fx["gen_eval"])
    sums_close({k: sd[k] for k in gsd if "running" in k or "tracked" in k}, fx["gen_bn_after"])
    fake = rows.view(B, T, 69).permute(0, 2, 1).contiguous()
    d_params, _ = O.split_state(dsd)
    D = lambda x, a=None: O.p3_critic(d_params, x, a, 25, activ, ablated)
    a_in = None if ablated else audio_c
    close(D(real_c, a_in), fx["score_real"])
    close(D(fake.detach(), a_in), fx["score_fake"])
    torch.manual_seed(9)
    alpha = torch.rand(B, 1)
    close(alpha, fx["alpha"], 0)
    gp, _, _ = O.gradient_penalty(D, real_c, fake, alpha, None if ablated else audio_c.clone(), True, False)
    assert abs(gp.item() - fx["gp"]) < 2e-5 * max(1.0, abs(fx["gp"]))
    err_critic = D(fake.detach(), a_in).mean() - D(real_c, a_in).mean() + 10 * gp
    assert abs(err_critic.item() - fx["err_critic"]) < 1e-4 * max(1.0, abs(fx["err_critic"]))
    norms_close(O.grads_of(err_critic, d_params), param_names(dsd), fx["critic_grad_norms"])
    fake_g = rows.view(B, T, 69).permute(0, 2, 1)
    l1 = (real_c - fake_g).abs().mean()
    err_gen = D(real_c, a_in).mean() - D(fake_g, a_in).mean() + 1.0 * l1 + 0.0 * O.tv_loss(fake_g)
    assert abs(l1.item() - fx["err_l1"]) < 2e-5 and abs(err_gen.item() - fx["err_gen"]) < 1e-4
    norms_close(O.grads_of(err_gen, g_params), param_names(gsd), fx["gen_grad_norms"])


@pytest.mark.parametrize("case", [("default", "id", False), ("wavegan", "id", False), ("unet", "id", True)],
                         ids=lambda c: "%s_%s_%s" % c)
def test_p3_trace(case):
    enc, activ, ablated = case
    fx = load(p3_name(enc, activ, ablated, 120))
    gsd, dsd = filled(fx, "gen", 5000), filled(fx, "critic", 6000)
    B, T = 2, 120
    real, aud = P.poses(B, T, seed=41), P.audio(B, T, seed=42)
    cfg = O.P3Config(enc_type=enc, activ=activ, ablated=ablated)
    tr, g_out, d_out = O.p3_train_iterations(gsd, dsd, cfg, real, aud, P.slices(aud), 8, 10)
    for k in ("loss_critic", "gp", "w_dist", "loss_gen", "err_l1"):
        close(np.array(tr[k]), fx["trace_" + k], 1e-3, 2e-2)
    close(np.array(tr["loss_critic"][:1]), fx["trace_loss_critic"][:1], 2e-4)
    assert tr["g_step"] == [0] * 7 + [1]
    sums_close({k: g_out[k] for k in gsd}, fx["gen_final_sum"], adam_lr=2e-4, adam_steps=1)
    sums_close({k: d_out[k] for k in dsd}, fx["critic_final_sum"], adam_lr=2e-4, adam_steps=8)
