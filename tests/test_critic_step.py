"""The hand-scheduled critic iteration (music2dance_amd/critic_step.py) against the autograd path it replaces
(losses.gradient_penalty + critic.score_pair + backward: the path the reference fixtures pin, test_product_parity.py):
same losses, and every parameter gradient element for element. Kernel layers as elsewhere: `cpu-fake` runs the
schedule's host logic here, `hip` the real kernels on the GPU box (incl. BASELINE's batch 64)."""
import pytest
import torch

from music2dance_amd import kernels
from music2dance_amd.critic_step import CriticStep
from music2dance_amd.losses import gradient_penalty
from music2dance_amd.phase2.archis import default as p2
from music2dance_amd.phase3.archis import default as p3


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


def _inputs(B, T, dev, audio=True, seed=3):
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(B, T, 69, generator=g).to(dev)
    fake_rows = (torch.rand(B * T, 69, generator=g) * 1.5 - 0.2).to(dev)
    alpha = torch.rand(B, 1, generator=g).to(dev)
    aud = (0.1 * torch.randn(B, 1, 640 * T, generator=g)).to(dev) if audio else None
    return real, fake_rows, alpha, aud


def _autograd(critic, real, fake_rows, alpha, audio, gamma, lp):
    """the engines' round-2 critic iteration (engine._critic_passes)"""
    B, T = real.shape[0], real.shape[1]
    for p in critic.parameters():
        p.grad = None
    real_c = real.permute(0, 2, 1).contiguous()
    fake = fake_rows.view(B, T, 69).permute(0, 2, 1).contiguous()
    if audio is None:
        gp = gradient_penalty(critic, B, real_c, fake, is_seq=True, lp=lp, device=real.device, alpha=alpha)
        s_real, s_fake = critic.score_pair(real_c, fake)
    else:
        a = audio.clone()
        with critic.shared_audio():
            gp = gradient_penalty(critic, B, real_c, fake, a, is_seq=True, lp=lp, device=real.device, alpha=alpha)
            s_real, s_fake = critic.score_pair(real_c, fake, a)
    w = s_fake.mean() - s_real.mean()
    loss = w + gamma * gp
    loss.backward()
    grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in critic.named_parameters()}
    return {"loss_critic": loss.detach(), "gp": gp.detach(), "w_dist": w.detach()}, grads


def _compare(critic, real, fake_rows, alpha, audio, gamma, lp, rtol, flips=False):
    want, ref = _autograd(critic, real, fake_rows, alpha, audio, gamma, lp)
    step = CriticStep(critic, gamma, lp=lp)
    got = step.run(real, fake_rows, None if audio is None else audio.clone(), alpha)
    if real.is_cuda:
        torch.cuda.synchronize()
    for k in want:
        a, b = float(got[k]), float(want[k])
        assert abs(a - b) <= rtol * max(1.0, abs(b)), (k, a, b)
    worst = 0.0
    for n, p in critic.named_parameters():
        assert (p.grad is None) == (ref[n] is None), n
        if ref[n] is None:
            continue
        assert p.grad.shape == ref[n].shape, n
        scale = ref[n].abs().max().item()
        diff = (p.grad - ref[n]).abs()
        err = diff.max().item()
        worst = max(worst, err / max(scale, 1e-30))
        if flips:
            # full size: the two fp32 schedules do not share every ReLU mask (a pre-activation within rounding of zero
            # flips; see tests/test_gpu_full_size_parity.py::_norms_close): a few elements may move by O(1e-3 .. 1e-2)
            assert err <= 3e-2 * scale + 1e-9, "%s: max |manual - autograd| %.3e vs max |autograd| %.3e" % (n, err, scale)
            n_off = int((diff > rtol * scale + 1e-9).sum())
            assert n_off <= max(2, int(0.02 * diff.numel())), "%s: %d of %d elements off by > %.0e" % (n, n_off, diff.numel(), rtol)
        else:
            assert err <= rtol * scale + 1e-9, "%s: max |manual - autograd| %.3e vs max |autograd| %.3e" % (n, err, scale)
    return worst


@pytest.mark.parametrize("activ", ["id", "relu"])
def test_phase3_two_branch_critic(dev, activ):
    torch.manual_seed(0)
    critic = p3.SequenceDiscriminator(69, 16, 12, 120, init_ker=25, activ=activ, device="cpu").to(dev)
    real, fake_rows, alpha, audio = _inputs(2, 120, dev)
    _compare(critic, real, fake_rows, alpha, audio, 10.0, False, 2e-4 if dev.type == "cuda" else 1e-5)


def test_phase3_ablated_critic(dev):
    torch.manual_seed(1)
    critic = p3.AblatedSequenceDiscriminator(69, 16, 12, 40, init_ker=25, activ="id", device="cpu").to(dev)
    real, fake_rows, alpha, _ = _inputs(3, 40, dev, audio=False)
    _compare(critic, real, fake_rows, alpha, None, 10.0, False, 2e-4 if dev.type == "cuda" else 1e-5)


@pytest.mark.parametrize("lp", [True, False])
def test_phase2_critic(dev, lp):
    torch.manual_seed(2)
    critic = p2.SequenceDiscriminator(69, 16, 30, 25, 3, "cpu").to(dev)
    real, fake_rows, alpha, _ = _inputs(4, 30, dev, audio=False)
    _compare(critic, real, fake_rows, alpha, None, 10.0, lp, 2e-4 if dev.type == "cuda" else 1e-5)


def test_tanh_heads_are_supported():
    assert CriticStep.supports(p3.AblatedSequenceDiscriminator(69, 8, 6, 20, activ="tanh", device="cpu"))
    assert CriticStep.supports(p3.AblatedSequenceDiscriminator(69, 8, 6, 20, activ="relu", device="cpu"))


def test_phase3_two_branch_critic_with_tanh_heads(dev):
    """`activ: tanh` (phase3/configs/tanh.yaml): the penalty's double backward through tanh - the tangent times
    (1 - e^2) plus the tanh'' cotangent chain on the 4B-row layout - against autograd through ops.tanh."""
    torch.manual_seed(0)
    critic = p3.SequenceDiscriminator(69, 16, 12, 120, init_ker=25, activ="tanh", device="cpu").to(dev)
    real, fake_rows, alpha, audio = _inputs(2, 120, dev)
    _compare(critic, real, fake_rows, alpha, audio, 10.0, False, 2e-4 if dev.type == "cuda" else 1e-5)


def test_phase3_ablated_critic_with_tanh_head(dev):
    torch.manual_seed(1)
    critic = p3.AblatedSequenceDiscriminator(69, 16, 12, 40, init_ker=25, activ="tanh", device="cpu").to(dev)
    real, fake_rows, alpha, _ = _inputs(3, 40, dev, audio=False)
    _compare(critic, real, fake_rows, alpha, None, 10.0, False, 2e-4 if dev.type == "cuda" else 1e-5)


@pytest.mark.gpu
def test_full_size_batch_64_matches_autograd():
    """BASELINE configs[2]'s critic at its size (B = 64, 120 frames): the launch plans the bench runs."""
    import bench
    dev = torch.device("cuda:0")
    _, critic = bench.build_models(dev, 120)
    real, fake_rows, alpha, audio = _inputs(64, 120, dev, seed=9)
    with kernels.impl().weight_cache():
        worst = _compare(critic, real, fake_rows, alpha, audio, 10.0, False, 1e-3, flips=True)
    print("manual vs autograd at B=64: worst element error relative to the tensor's largest: %.2e" % worst)


@pytest.mark.gpu
def test_full_size_batch_64_tanh_critic_matches_autograd():
    """The tanh critic at BASELINE configs[2]'s size (B = 64, 120 frames)."""
    import bench
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    critic = p3.SequenceDiscriminator(69, 128, 100, 120, init_ker=25, activ="tanh", device="cpu").to(dev)
    real, fake_rows, alpha, audio = _inputs(64, 120, dev, seed=9)
    with kernels.impl().weight_cache():
        worst = _compare(critic, real, fake_rows, alpha, audio, 10.0, False, 1e-3, flips=True)
    print("tanh critic, manual vs autograd at B=64: worst element error relative to the tensor's largest: %.2e" % worst)


@pytest.mark.gpu
def test_phase2_captured_graphs_equal_eager():
    """Phase2Engine.enable_graphs(): nine loop bodies (one generator iteration) replayed from captured graphs leave the
    same parameters and losses as the eager engine with the same host draws (BASELINE configs[1] shapes, batch 8)."""
    from music2dance_amd.engine import Phase2Engine
    dev = torch.device("cuda:0")
    cfg = {"lr_gen": 5e-4, "lr_critic": 5e-4, "n_critic_steps": 8, "gamma": 10, "eta": 50, "input_vector_size": 50}
    real = torch.rand(8, 120, 69, generator=torch.Generator().manual_seed(4)).to(dev)
    sigs = []
    for graphs in (False, True):
        torch.manual_seed(0)
        gen = p2.SequenceGenerator(50, 50, 256, 69, 2, 3, dev)
        critic = p2.SequenceDiscriminator(69, 128, 120, 25, 3, dev)
        eng = Phase2Engine(gen, critic, cfg, data_parallel=False)
        if graphs:
            eng.enable_graphs()
        torch.manual_seed(21)
        outs = []
        for _ in range(9):
            out = eng.train_step(real)
            outs.append([float(out[k]) for k in ("loss_critic", "gp", "w_dist")])
        eng.flush()
        torch.cuda.synchronize()
        sigs.append((outs, float(out["loss_gen"]) if "loss_gen" in out else None,
                     [p.detach().double().sum().item() for p in list(critic.parameters()) + list(gen.parameters())]))
    (o0, g0, s0), (o1, g1, s1) = sigs
    for a, b in zip(sum(o0, []) + s0, sum(o1, []) + s1):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (a, b)


@pytest.mark.gpu
def test_phase2_graph_replays_without_host_syncs_equal_eager():
    """48 loop bodies with NO host read between them: the generator-forward graph of body i + 1 then really runs on the
    generator stream underneath the critic graph of body i (a float() per step, as in the test above, serialises the two
    streams). Both graphs count split-K arrivals and BatchNorm / channel sums in zero-kept scratch; captured on torch's
    one capture stream they shared it (round-3 advice: wrong sums, scratch left dirty) - each graph now owns its scratch
    (kernels.private_scratch). Compared with the eager engine under the same host draws, at lr 5e-5 (phase 2 at its
    config's 5e-4 amplifies rounding, see test_product_parity.py)."""
    from music2dance_amd.engine import Phase2Engine
    dev = torch.device("cuda:0")
    cfg = {"lr_gen": 5e-5, "lr_critic": 5e-5, "n_critic_steps": 8, "gamma": 10, "eta": 50, "input_vector_size": 50}
    real = torch.rand(8, 120, 69, generator=torch.Generator().manual_seed(4)).to(dev)
    ready = torch.cuda.current_stream(dev).record_event()
    sigs = []
    for graphs in (False, True):
        torch.manual_seed(0)
        gen = p2.SequenceGenerator(50, 50, 256, 69, 2, 3, dev)
        critic = p2.SequenceDiscriminator(69, 128, 120, 25, 3, dev)
        eng = Phase2Engine(gen, critic, cfg, data_parallel=False)
        if graphs:
            eng.enable_graphs()
        torch.manual_seed(21)
        outs = []
        for _ in range(48):
            out = eng.train_step(real, inputs_ready=ready)
            outs.append(torch.stack([out[k].detach().clone() for k in ("loss_critic", "gp", "w_dist")]))
        eng.flush()
        torch.cuda.synchronize()
        sigs.append((torch.stack(outs).double().cpu(),
                     [p.detach().double().sum().item() for p in list(critic.parameters()) + list(gen.parameters())]))
    (o0, s0), (o1, s1) = sigs
    assert torch.isfinite(o1).all()
    worst = ((o0 - o1).abs() / o0.abs().clamp_min(1.0)).max().item()
    assert worst <= 1e-3, "losses over 48 bodies: graphs vs eager differ by %.3e" % worst
    for a, b in zip(s0, s1):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(a)), (a, b)


@pytest.mark.gpu
def test_phase1_captured_graphs_equal_eager():
    """Phase1Engine.enable_graphs() (round 3): every host draw of a loop body - generator noise, interpolation weights,
    the dropout masks of both networks - goes through a DrawTape and is made again, in the same order on the same
    generator, before each replay; six loop bodies (one generator iteration) then leave the same losses and parameters
    as the eager engine (BASELINE configs[0] shapes)."""
    from music2dance_amd.engine import Phase1Engine
    from music2dance_amd.phase1.archis.residual import Discriminator as D1, Generator as G1
    dev = torch.device("cuda:0")
    cfg = {"lr_gen": 1e-4, "lr_critic": 1e-4, "n_critic_steps": 5, "gamma": 10, "latent_vector_size": 10}
    real = torch.rand(64, 23, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    sigs = []
    for graphs in (False, True):
        torch.manual_seed(0)
        gen = G1(10, 128, 69, 1).to(dev)
        critic = D1(69, 128, 1).to(dev)
        eng = Phase1Engine(gen, critic, cfg, data_parallel=False)
        if graphs:
            eng.enable_graphs()
        torch.manual_seed(21)
        outs = []
        for _ in range(6):
            out = eng.train_step(real)
            outs.append([float(out[k]) for k in ("loss_critic", "gp", "w_dist")])
        eng.flush()
        torch.cuda.synchronize()
        sigs.append((outs, float(out["loss_gen"]) if "loss_gen" in out else None,
                     [p.detach().double().sum().item() for p in list(critic.parameters()) + list(gen.parameters())]))
    (o0, g0, s0), (o1, g1, s1) = sigs
    assert g0 is None and g1 is None or abs(g0 - g1) <= 2e-5 * max(1.0, abs(g0))
    for a, b in zip(sum(o0, []) + s0, sum(o1, []) + s1):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (a, b)
