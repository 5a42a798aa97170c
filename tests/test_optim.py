"""music2dance_amd.optim.Adam (one multi-tensor launch per step, include/m2d.h: m2d_adam_multi) against torch.optim.Adam:
same trajectories, same state_dict layout, parameters without a gradient skipped, conv-weight images refreshed in the
same pass. `cpu-fake` checks the host logic here, `hip` the kernel on the GPU box."""
import pytest
import torch

from music2dance_amd import kernels
from music2dance_amd.optim import Adam


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


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(128, 69, 25), (128,), (128, 128, 7), (128,), (100, 12800), (1, 128), (3,), (5000, 33), (64, 32, 4)]
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]


def test_trajectory_equals_torch_adam(dev):
    ours, ref = _params(dev), _params(dev)
    dead = 6  # a parameter that never gets a gradient (LinearBlock.fc1 / bn1 of the reference): skipped, no state
    opt = Adam(ours, lr=2e-4)
    opt_ref = torch.optim.Adam(ref, lr=2e-4)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for i, (a, b) in enumerate(zip(ours, ref)):
            if i == dead or (i == 3 and step < 2):  # (index 3 gets its first gradient at step 2: own step count)
                a.grad = b.grad = None
                continue
            gr = (torch.randn(a.shape, generator=g) * (10.0 ** (i % 3 - 1))).to(dev)
            a.grad, b.grad = gr.clone(), gr.clone()
        opt.step()
        opt_ref.step()
    for i, (a, b) in enumerate(zip(ours, ref)):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (i, (a - b).abs().max().item())
    assert ours[dead] not in opt.state
    sd, sd_ref = opt.state_dict(), opt_ref.state_dict()
    assert sorted(sd["state"].keys()) == sorted(sd_ref["state"].keys())
    for k in sd["state"]:
        assert sorted(sd["state"][k].keys()) == ["exp_avg", "exp_avg_sq", "step"]
        assert float(sd["state"][k]["step"]) == float(sd_ref["state"][k]["step"])
        m, m_ref = sd["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"]
        assert (m - m_ref).abs().max().item() <= 1e-6 * m_ref.abs().max().item()
    # a torch.optim.Adam can resume from our state (the checkpoint surface)
    again = torch.optim.Adam(ref, lr=2e-4)
    again.load_state_dict(sd)


def test_lr_schedulers_drive_the_group_learning_rate(dev):
    ps = _params(dev)[:2]
    opt = Adam(ps, lr=5e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1, 2], gamma=0.8)   # phase2/train.py:88-89
    before = [p.detach().clone() for p in ps]
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step(); sched.step()
    d1 = (before[0] - ps[0].detach()).abs().max().item()
    assert abs(d1 - 5e-4) <= 1e-6   # first Adam step = lr * sign(g)
    assert abs(opt.param_groups[0]["lr"] - 4e-4) <= 1e-12


@pytest.mark.gpu
def test_packed_conv_images_are_rewritten_by_the_step():
    """A conv weight with live packed images in the weight cache: after the step the cached (w_fwd, w_bwd) equal a fresh
    m2d_conv1d_pack_weights of the updated weight, with no pack launch; a skip flag voids the whole step."""
    dev = torch.device("cuda:0")
    K = kernels.impl()
    w = torch.randn(128, 64, 25, device=dev).requires_grad_(True)
    b = torch.randn(128, device=dev).requires_grad_(True)
    opt = Adam([w, b], lr=1e-2)
    with K.weight_cache():
        wf0, wb0 = K.packed_weights(w)
        w.grad, b.grad = torch.randn_like(w), torch.randn_like(b)
        before = w.detach().clone()
        launches = K.pack_launches
        opt.step()
        K.invalidate_packed([w])                      # what the engines call after a step: must keep the fresh images
        wf1, wb1 = K.packed_weights(w)
        assert K.pack_launches == launches and wf1 is wf0 and wb1 is wb0
        assert not torch.equal(before, w.detach())
        assert torch.equal(wf1, w.detach().permute(1, 2, 0).contiguous())
        assert torch.equal(wb1, w.detach().permute(0, 2, 1).contiguous())
        opt.skip_flag = torch.ones((), device=dev)
        held = w.detach().clone()
        w.grad = torch.randn_like(w)
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(held, w.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 32, 25), (100, 36, 7), (128, 69, 25), (128, 128, 7), (512, 256, 25), (20, 16, 3)])
def test_packed_images_after_the_step_for_every_tile_shape(shape):
    """Round 5: the step of a conv weight with live packed images goes through 8 x 256 LDS tiles (rows that are not a
    multiple of 4 floats long: the one-element path). Parameters and moments bit-equal to torch.optim.Adam(foreach=False),
    both images equal to a fresh pack of the updated weight - over three steps, with output-channel counts that are not
    multiples of 8 and rows that end inside a tile."""
    dev = torch.device("cuda:0")
    K = kernels.impl()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    w_ref = w.detach().clone().requires_grad_(True)
    opt, opt_ref = Adam([w], lr=1e-3), torch.optim.Adam([w_ref], lr=1e-3, foreach=False)
    with K.weight_cache():
        K.packed_weights(w)
        for _ in range(3):
            gr = torch.randn(shape, generator=g).to(dev)
            w.grad, w_ref.grad = gr.clone(), gr.clone()
            opt.step()
            opt_ref.step()
            K.invalidate_packed([w])
            wf, wb = K.packed_weights(w)
            assert torch.allclose(w.detach(), w_ref.detach(), rtol=2e-6, atol=2e-7)
            assert torch.equal(wf, w.detach().permute(1, 2, 0).contiguous())
            assert torch.equal(wb, w.detach().permute(0, 2, 1).contiguous())
    st, st_ref = opt.state[w], opt_ref.state[w_ref]
    # (moments: relative to the tensor's largest element; 1 - beta2 is formed in fp32 here and in fp64 by torch: 1.3e-5)
    for key, tol in (("exp_avg", 1e-6), ("exp_avg_sq", 3e-5)):
        a, b = st[key], st_ref[key]
        assert (a - b).abs().max().item() <= tol * b.abs().max().item(), key
