"""The train-script surface of the path (reference phase3/train.py:239-276,
phase2/train.py:88-89,179-180): validation loop + `l1_loss_val`, checkpoint file names and
round trip, strict loading of reference-layout state_dicts, MultiStepLR values.

Kernel layers as in test_product_parity.py: `cpu-fake` exercises the host logic here, `hip`
runs the same scripts on the MI355X box.
"""
import glob
import os

import numpy as np
import pytest
import torch
import yaml

from music2dance_amd import kernels, runner
from music2dance_amd.engine import Phase2Engine
from music2dance_amd.phase2.archis import default as p2
from music2dance_amd.phase3.archis import default as p3
from tests.golden import patterns as P

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "music2dance_amd")


@pytest.fixture(params=[pytest.param("cpu-fake"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request, monkeypatch):
    if request.param == "hip":
        assert kernels.impl().name == "hip"
        yield torch.device("cuda:0")
    else:
        from tests.fake_backend import FakeKernels
        prev = kernels.set_impl(FakeKernels())
        monkeypatch.setattr(runner, "pick_device", lambda idx: torch.device("cpu"))
        try:
            yield torch.device("cpu")
        finally:
            kernels.set_impl(prev)


def _cfg(tmp_path, src, **over):
    cfg = yaml.safe_load(open(os.path.join(PKG, src)))
    cfg.update(over)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    return str(path)


def test_phase3_train_script_validation_and_checkpoints(dev, tmp_path, monkeypatch):
    """Two epochs of 2 critic + 1 generator iterations: `l1_loss_val` is produced by an eval-mode
    pass after every epoch (n_valid_steps = 1), the generator returns to train mode, and the
    checkpoint written under the reference's file name restores identical outputs."""
    from music2dance_amd.phase3 import train as T
    monkeypatch.chdir(tmp_path)
    cfg = _cfg(tmp_path, "phase3/configs/ablated.yaml", batch_size=2, num_train=4, num_epochs=2, n_critic_steps=2)
    eng = T.main(["-c", cfg, "-d", "0", "-n", "t", "--synthetic"])
    assert eng.total_iterations == 4
    assert eng.gen.training
    logs = glob.glob(str(tmp_path / "runs" / "*_t"))
    assert len(logs) == 1 and os.path.isdir(logs[0] + "/models") and os.path.isdir(logs[0] + "/samples")
    assert os.path.isfile(logs[0] + "/model_gen.txt") and os.path.isfile(logs[0] + "/model_critic.txt")
    val = float(T.LAST_LOG.last["l1_loss_val"])
    assert np.isfinite(val) and val > 0
    for tag in ("loss_critic", "loss_gen", "gp", "w_dist", "l1_loss_train", "l1_loss_val"):
        assert tag in T.LAST_LOG.last, tag

    # checkpoint round trip under the reference's names (phase3/train.py:267-276)
    path = logs[0] + "/models/gpgen_100.pt"
    runner.save_state(eng.gen, path)
    runner.save_state(eng.critic, logs[0] + "/models/gpcritic_5000.pt")
    torch.manual_seed(3)
    fresh = p3.SequenceGenerator(3200, 250, 250, 256, 69, 10, 2, 3, "default", "id", dev)
    fresh.load_state_dict(torch.load(path, map_location=dev), strict=True)
    fresh_c = p3.AblatedSequenceDiscriminator(69, 128, 100, 120, init_ker=25, activ="id", device=dev)
    fresh_c.load_state_dict(torch.load(logs[0] + "/models/gpcritic_5000.pt", map_location=dev), strict=True)
    slices = (0.1 * torch.randn(2, 120, 3200, generator=torch.Generator().manual_seed(1))).to(dev)
    noise = torch.randn(2, 120, 10, generator=torch.Generator().manual_seed(2)).to(dev)
    eng.gen.eval(), fresh.eval()
    with torch.no_grad():
        a = eng.gen(slices, [120, 120], noise)
        b = fresh(slices, [120, 120], noise)
        x = torch.rand(2, 69, 120, generator=torch.Generator().manual_seed(4)).to(dev)
        sa, sb = eng.critic(x), fresh_c(x)
    assert torch.equal(a, b) and torch.equal(sa, sb)


@pytest.mark.parametrize("fixture,enc,ablated", [("p3_default_id_full", "default", False),
                                                 ("p3_unet_id_abl", "unet", True),
                                                 ("p3_wavegan_id_full", "wavegan", False)])
def test_reference_layout_state_dicts_load_strict(fixture, enc, ablated):
    """A state_dict with exactly the reference's keys and shapes (recorded from the imported
    reference modules by tests/golden/make_golden.py) loads with strict=True."""
    fx = np.load(os.path.join(HERE, "golden", fixture + ".npz"))
    gen = p3.SequenceGenerator(P.WINDOW, 250, 250, 256, 69, 10, 2, 3, enc, "id", "cpu")
    cls = p3.AblatedSequenceDiscriminator if ablated else p3.SequenceDiscriminator
    critic = cls(69, 128, 100, 120, init_ker=25, activ="id", device="cpu")
    for module, which in ((gen, "gen"), (critic, "critic")):
        ref_sd = {}
        for k, shp in zip(fx[which + "_keys"], fx[which + "_shapes"]):
            shape = tuple(int(d) for d in str(shp).split(",")) if str(shp) else ()
            dt = torch.long if str(k).endswith("num_batches_tracked") else torch.float32
            ref_sd[str(k)] = torch.zeros(shape, dtype=dt)
        missing, unexpected = module.load_state_dict(ref_sd, strict=True)
        assert not missing and not unexpected


def test_phase2_train_script_runs(dev, tmp_path, monkeypatch):
    from music2dance_amd.phase2 import train as T
    monkeypatch.chdir(tmp_path)
    cfg = _cfg(tmp_path, "phase2/configs/default.yaml", batch_size=2, num_train=4, num_epochs=1, n_critic_steps=2)
    eng = T.main(["-c", cfg, "-d", "0", "-n", "t2", "-f", "wgangp", "--synthetic", "--no-run-dir"])
    assert eng.total_iterations == 2 and "loss_gen" in eng.last
    with pytest.raises(ValueError, match="Please state existing framework"):
        T.main(["-c", cfg, "-d", "0", "-n", "t2", "-f", "nope", "--synthetic"])


def test_phase1_train_script_runs(dev, tmp_path, monkeypatch):
    from music2dance_amd.phase1 import train_wgan_gp as T
    monkeypatch.chdir(tmp_path)
    cfg = _cfg(tmp_path, "phase1/configs/b2l50s32.yaml", batch_size=8, num_train=40, num_epochs=1)
    eng = T.main(["-c", cfg, "-d", "0", "-n", "t1", "--synthetic", "--no-run-dir"])
    assert eng.total_iterations == 5 and "loss_gen" in eng.last


# ------------------------------------------------------------------------------ dataset path
def test_train_scripts_run_on_a_dataset_folder(dev, tmp_path, monkeypatch):
    """Without --synthetic the scripts go through music2dance_amd.data (the reference's StickDataset /
    SequenceDataset / collate_fn / seeded split + class-balanced samplers) on a folder in the dataset's on-disk
    format; a missing folder exits with a message instead of a traceback."""
    from music2dance_amd.data import write_synthetic_dataset
    from music2dance_amd.phase1 import train_wgan_gp as T1
    from music2dance_amd.phase2 import train as T2
    from music2dance_amd.phase3 import train as T3
    monkeypatch.chdir(tmp_path)
    folder = write_synthetic_dataset(str(tmp_path / "ds"), n_takes=10, seconds=6, seed=1)
    c3 = _cfg(tmp_path, "phase3/configs/ablated.yaml", batch_size=2, num_epochs=1, n_critic_steps=2, folder=folder)
    eng = T3.main(["-c", c3, "-d", "0", "-n", "d3"])
    assert eng.total_iterations == 4 and "loss_gen" in eng.last_full  # 8 training takes / batch 2
    assert torch.isfinite(T3.LAST_LOG.last["l1_loss_val"])
    run = glob.glob(str(tmp_path / "runs" / "*_d3"))[0]
    split = yaml.safe_load(open(run + "/trainvaltest_samples.json"))
    assert sorted(len(v) for v in split.values()) == [1, 1, 8]
    c2 = _cfg(tmp_path, "phase2/configs/default.yaml", batch_size=2, num_train=6, num_epochs=1, n_critic_steps=2)
    eng = T2.main(["-c", c2, "-d", "0", "-n", "d2", "--no-run-dir", "--folder", folder])
    assert eng.total_iterations == 3
    c1 = _cfg(tmp_path, "phase1/configs/b2l50s32.yaml", batch_size=8, num_train=40, num_epochs=1)
    eng = T1.main(["-c", c1, "-d", "0", "-n", "d1", "--no-run-dir", "--folder", folder])
    assert eng.total_iterations == 5
    c3 = _cfg(tmp_path, "phase3/configs/ablated.yaml", batch_size=2, num_epochs=1, n_critic_steps=2)  # (same file name)
    with pytest.raises(SystemExit, match="dataset folder"):
        T3.main(["-c", c3, "-d", "0", "-n", "d3x", "--no-run-dir", "--folder", str(tmp_path / "nope")])


@pytest.mark.gpu
def test_resident_and_host_loaders_train_identically(tmp_path, monkeypatch):
    """The train scripts' default on a GPU - dataset in HBM, one gather per batch on the copy stream
    (data.ResidentLoader) - against --host-loader (torch DataLoader + collate on the host, as the reference does):
    same batches, so the same parameters after a few loop bodies of each phase."""
    assert kernels.impl().name == "hip"
    from music2dance_amd.data import write_synthetic_dataset
    from music2dance_amd.phase1 import train_wgan_gp as T1
    from music2dance_amd.phase2 import train as T2
    from music2dance_amd.phase3 import train as T3
    monkeypatch.chdir(tmp_path)
    folder = write_synthetic_dataset(str(tmp_path / "ds"), n_takes=12, seconds=6, seed=2)
    faults = kernels.impl().async_faults
    runs = {}
    for extra in ([], ["--host-loader"]):
        c3 = _cfg(tmp_path, "phase3/configs/default.yaml", batch_size=4, num_epochs=2, n_critic_steps=2, folder=folder)
        e3 = T3.main(["-c", c3, "-d", "0", "-n", "r3", "--no-run-dir"] + extra)
        c2 = _cfg(tmp_path, "phase2/configs/default.yaml", batch_size=4, num_train=10, num_epochs=2, n_critic_steps=2)
        e2 = T2.main(["-c", c2, "-d", "0", "-n", "r2", "--no-run-dir", "--folder", folder] + extra)
        c1 = _cfg(tmp_path, "phase1/configs/b2l50s32.yaml", batch_size=8, num_train=40, num_epochs=1)
        e1 = T1.main(["-c", c1, "-d", "0", "-n", "r1", "--no-run-dir", "--folder", folder] + extra)
        runs[bool(extra)] = [(e.total_iterations, [p.detach().clone() for m in (e.gen, e.critic) for p in m.parameters()])
                             for e in (e3, e2, e1)]
    assert kernels.impl().async_faults == faults, "a recurrent launch timed out during the runs (recovered, steps voided)"
    for phase, ((na, pa), (nb, pb)) in zip((3, 2, 1), zip(runs[False], runs[True])):
        assert na == nb and na >= 4
        bad = [(i, float((a - b).abs().max())) for i, (a, b) in enumerate(zip(pa, pb)) if not torch.equal(a, b)]
        assert not bad, "phase %d: %d of %d tensors differ: %s" % (phase, len(bad), len(pa), bad[:6])


def test_scalar_log_writes_every_value_without_blocking_the_loop(dev):
    """runner.ScalarLog under a writer: values logged as device tensors arrive with their tags and steps, in order,
    although `scalars()` never waits for the device (pinned copies + events; flush() at the end writes the rest)."""
    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, value, step))

    w = Writer()
    log = runner.ScalarLog(None, every=2, writer=w)
    want = []
    for step in range(1, 9):
        vals = {"loss_gen": torch.tensor(float(step), device=dev) * 0.5, "gp": float(step) + 0.25}
        log.scalars(vals, step)
        if step % 2 == 0:
            want += [("loss_gen", step * 0.5, step), ("gp", step + 0.25, step)]
    log.scalars({"l1_loss_val": torch.tensor(3.0, device=dev)}, 9, force=True)
    want.append(("l1_loss_val", 3.0, 9))
    log.flush()
    assert w.rows == want and float(log.last["l1_loss_val"]) == 3.0


# ------------------------------------------------------------------------------ MultiStepLR
def _lr_closed_form(lr0, gen_iters, milestones=(10000, 35000, 50000), gamma=0.8):
    return lr0 * gamma ** sum(1 for m in milestones if gen_iters >= m)


def test_phase2_multistep_lr_matches_reference_schedule():
    """phase2/train.py:88-89: MultiStepLR(milestones=[10000, 35000, 50000], gamma=0.8) on both
    optimisers, stepped once per GENERATOR iteration (:179-180)."""
    from tests.fake_backend import FakeKernels
    prev = kernels.set_impl(FakeKernels())
    try:
        torch.manual_seed(0)
        gen = p2.SequenceGenerator(8, 8, 16, 69, 1, 1, "cpu")
        critic = p2.SequenceDiscriminator(69, 8, 24, 5, 1, "cpu")
        cfg = {"lr_gen": 1e-4, "lr_critic": 2e-4, "n_critic_steps": 2, "gamma": 10, "eta": 50, "input_vector_size": 8}
        eng = Phase2Engine(gen, critic, cfg, data_parallel=False)
        for sch in (eng.scheduler_gen, eng.scheduler_critic):
            assert sorted(sch.milestones) == [10000, 35000, 50000] and sch.gamma == 0.8
        # through the real loop: the schedulers advance on generator iterations only
        real = torch.rand(2, 24, 69)
        for _ in range(5):
            eng.train_step(real)
        assert eng.scheduler_gen.last_epoch == 2 and eng.scheduler_critic.last_epoch == 2
        # far side of every milestone: stepping the scheduler objects the engine owns
        marks = {9999, 10000, 34999, 35000, 49999, 50000, 50001}
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for n in range(eng.scheduler_gen.last_epoch + 1, 50002):
                eng.scheduler_gen.step(), eng.scheduler_critic.step()
                if n in marks:
                    assert eng.optim_gen.param_groups[0]["lr"] == pytest.approx(_lr_closed_form(1e-4, n), rel=1e-12)
                    assert eng.optim_critic.param_groups[0]["lr"] == pytest.approx(_lr_closed_form(2e-4, n), rel=1e-12)
    finally:
        kernels.set_impl(prev)
