"""Dataset side of the train scripts: music2dance_amd.data / losses.jerkiness against
OUTPUTS of the reference's own code (tests/golden/data.npz, written by make_golden.py::case_data from
utils.py:15-201,245-248,320-326, losses.py:85-89 and the split block of phase3/train.py:112-143).
Host logic runs here; the two HIP kernels behind it (m2d_jerk_mean_fwd, m2d_affine_cols) are checked on the GPU box."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "data.npz"), allow_pickle=False)


@pytest.fixture(params=["cpu-fake", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    from music2dance_amd import kernels
    if request.param == "hip":
        yield torch.device("cuda:0")
        return
    from tests.fake_backend import FakeKernels
    prev = kernels.set_impl(FakeKernels())
    yield torch.device("cpu")
    kernels.set_impl(prev)


def test_jerkiness_matches_reference(backend):
    from music2dance_amd.losses import jerkiness
    from oracle import m2d_oracle as O
    seq = torch.from_numpy(G["jerk_in"]).to(backend)
    want = float(G["jerk_f32"])
    assert abs(float(jerkiness(seq)) - want) <= 2e-6 * abs(want)
    # the generator's native (B, T, C) storage viewed as (B, C, T), as phase3/test.py:96 builds it
    perm = seq.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert abs(float(jerkiness(perm)) - float(G["jerk_perm"])) <= 2e-6 * abs(want)
    # float64 input (scaler.inverse_transform output): computed in fp32 here, fp64 in the reference
    assert abs(float(jerkiness(seq.double())) - float(G["jerk_f64"])) <= 2e-6 * abs(want)
    assert abs(float(O.jerkiness(torch.from_numpy(G["jerk_in"]))) - want) <= 1e-6 * abs(want)


def test_minmax_scaler_matches_sklearn_as_the_reference_uses_it(backend, tmp_path):
    from music2dance_amd.data import StickDataset
    path = str(tmp_path / "sk.npy")
    np.save(path, G["mm_in"])
    sd = StickDataset(path, resume=True, normalize="minmax")
    for key, got in (("mm_scale", sd.scaler.scale_), ("mm_min", sd.scaler.min_), ("mm_data_min", sd.scaler.data_min_),
                     ("mm_data_max", sd.scaler.data_max_), ("mm_scaled", sd.skeletons)):
        assert np.array_equal(G[key], got), key  # same fp64 operations in the same order: bit-equal
    assert np.array_equal(sd.scaler.inverse_transform(G["mm_inv_in"]), G["mm_inv"])
    assert np.array_equal(sd[3].numpy(), G["mm_item3"])
    # device forms (one kernel each): fp32 against the fp64 host result
    x = torch.from_numpy(G["mm_in"].reshape(50, 69)).float().to(backend)
    y = sd.scaler.transform_device(x)
    assert np.abs(y.cpu().numpy() - G["mm_scaled"].reshape(50, 69)).max() <= 1e-5
    back = sd.scaler.inverse_transform_device(y)
    assert np.abs(back.cpu().numpy() - G["mm_in"].reshape(50, 69)).max() <= 1e-4
    inv = sd.scaler.inverse_transform_device(torch.from_numpy(G["mm_inv_in"]).float().to(backend))
    assert np.abs(inv.cpu().numpy() - G["mm_inv"]).max() <= 1e-4
    z = x.clone()
    assert sd.scaler.transform_device(z, out=z) is z and torch.equal(z, y)  # in place


def test_folder_loaders_and_datasets_match_reference(tmp_path):
    from music2dance_amd import data as D
    folder = D.write_synthetic_dataset(str(tmp_path / "ds"), n_takes=6, seconds=6, seed=3)
    sticks = D.StickDataset(folder, normalize="minmax")
    cfg = {"audio_rate": 16000, "video_rate": 25, "seq_length": 4.8, "feat_size": 0.2}
    ds = D.SequenceDataset(folder, cfg, dance_types=["W", "C", "R", "T"], scaler=sticks.scaler, withaudio=True)
    ds.truncate()
    order = np.argsort([os.path.basename(d) for d in ds.dirs])
    assert [os.path.basename(ds.dirs[i]) for i in order] == list(G["ds_names"])
    assert [int(ds.labels[i]) for i in order] == list(G["ds_labels"])
    assert [len(ds.sequences[i]) for i in order] == list(G["ds_frames"])
    assert [len(ds.musics[i]) for i in order] == list(G["ds_samples"])
    assert np.allclose([np.sum(ds.sequences[i]) for i in order], G["ds_seq_sum"], rtol=1e-12, atol=0)
    assert np.allclose([np.abs(ds.sequences[i]).sum() for i in order], G["ds_seq_abs"], rtol=1e-12, atol=0)
    assert np.allclose([np.abs(ds.musics[i].astype(np.float64)).sum() for i in order], G["ds_music_abs"], rtol=1e-12)
    assert np.array_equal(sticks.scaler.data_min_, G["ds_scaler_min"]) and np.array_equal(sticks.scaler.data_max_, G["ds_scaler_max"])
    assert len(sticks) == int(G["ds_n_sticks"]) and abs(np.sum(sticks.skeletons) - float(G["ds_stick_sum"])) <= 1e-8
    assert [ds.stick_length, ds.audio_length, ds.ratio] == list(G["ds_lengths"])
    for j, i in enumerate(order):
        np.random.seed(100 + j)
        pose, music, label, d = ds[int(i)]
        assert pose.shape == (120, 23, 3) and music.shape == (76800,) and music.dtype == torch.float32
        assert abs(float(pose.double().sum()) - G["ds_crop_pose_sum"][j]) <= 1e-9
        assert abs(float(music.double().abs().sum()) - G["ds_crop_audio_abs"][j]) <= 1e-9
        np.random.seed(100 + j)
        assert D.get_positions(ds.sequences[int(i)], length=120)[0] == int(G["ds_crop_start"][j])
    # and the loaders built from it hand the engine batches of the shapes phase3/train.py:186-194 expects
    train_loader, val_loader, (tr, va, te) = D.make_loaders(ds, batch_size=3, logdir=None)
    assert len(tr) + len(va) + len(te) == 6
    real, lengths, audio, label, dirs = next(iter(train_loader))
    assert real.shape == (3, 120, 23, 3) and audio.shape == (3, 76800) and lengths == [120] * 3


def test_collate_fn_matches_reference():
    from music2dance_amd.data import collate_fn
    lens, labs = list(G["col_lens"]), list(G["col_labels"])
    batch = [(torch.from_numpy(G["col_seq%d" % j]), torch.from_numpy(G["col_mus%d" % j]),
              torch.from_numpy(np.asarray(labs[j])), "d%d" % j) for j in range(len(lens))]
    padded, lengths, musics, labels, dirs = collate_fn(list(batch))
    assert padded.dtype == torch.float32 and np.array_equal(padded.numpy(), G["col_padded"])
    assert lengths == list(G["col_lengths"]) and list(dirs) == list(G["col_dirs"])
    assert np.array_equal(musics.numpy(), G["col_musics"]) and np.array_equal(labels.numpy(), G["col_out_labels"])
    p2, l2, lab2, d2 = collate_fn([(b[0], b[2], b[3]) for b in batch], withaudio=False)
    assert abs(float(p2.double().sum()) - float(G["col2_padded_sum"])) <= 1e-9 and list(d2) == list(G["col2_dirs"])


def test_label_encoding_split_and_sampler_weights_match_reference():
    from music2dance_amd import data as D
    assert np.array_equal(D.one_hot_encode(list("WCRTTRCW")), G["onehot"])
    for n in (61, 8):
        tr, va, te = D.split_indices(n)
        assert tr == list(G["split%d_train" % n]) and va == list(G["split%d_val" % n]) and te == list(G["split%d_test" % n])
    tr = D.split_indices(61)[0]
    assert np.array_equal(D.class_balanced_weights(G["w61_labels"], tr), G["w61_train_weights"])


def test_batched_window_gather_equals_itemwise_crops_and_small_datasets_have_no_validation_loader(tmp_path):
    """Extensions of the rewritten dataset layer: sample_batch (one gather over the columnar store) returns what the
    item-by-item path + collate_fn returns under the same draws; an explicit crop generator; < 5 takes -> no hold-out."""
    from music2dance_amd import data as D
    folder = D.write_synthetic_dataset(str(tmp_path / "ds"), n_takes=4, seconds=6, seed=5)
    cfg = {"audio_rate": 16000, "video_rate": 25, "seq_length": 4.8, "feat_size": 0.2}
    ds = D.SequenceDataset(folder, cfg, withaudio=True)
    ds.truncate()
    idx = [2, 0, 3, 3]
    np.random.seed(7)
    want = D.collate_fn([ds[i] for i in idx])
    np.random.seed(7)
    got = ds.sample_batch(idx)
    assert torch.equal(got[0], want[0]) and got[1] == want[1] and torch.equal(got[2], want[2])
    assert torch.equal(got[3], want[3]) and tuple(got[4]) == tuple(want[4])
    # explicit generator: reproducible without touching numpy's global state
    a = D.SequenceDataset(folder, cfg, withaudio=False, crop_rng=np.random.RandomState(3))
    b = D.SequenceDataset(folder, cfg, withaudio=False, crop_rng=np.random.RandomState(3))
    state = np.random.get_state()[1].copy()
    assert all(torch.equal(a[i][0], b[i][0]) for i in (1, 1, 0))
    assert np.array_equal(state, np.random.get_state()[1])
    # per-take assignment re-packs the store
    first = np.array(ds.sequences[1])
    ds.sequences[1] = first[:150]
    assert len(ds.sequences[1]) == 150 and np.array_equal(ds.sequences[1], first[:150]) and len(ds.sequences[2]) > 0
    train_loader, val_loader, (tr, va, te) = D.make_loaders(ds, batch_size=2, logdir=None)
    assert val_loader is None and va == [] and len(tr) == 4 and next(iter(train_loader))[0].shape[0] == 2


def test_resident_loader_serves_the_dataloaders_batches(tmp_path):
    """data.ResidentLoader (dataset in device memory, one gather per batch) against torch's DataLoader over the same
    dataset / sampler / seeds: phase 3's class-balanced loaders (make_loaders), phase 2's SubsetRandomSampler with
    drop_last, phase 1's still poses - identical batches over two epochs, and the generators end in the same state."""
    from torch.utils.data import DataLoader, SubsetRandomSampler
    from music2dance_amd import data as D
    folder = D.write_synthetic_dataset(str(tmp_path / "ds"), n_takes=11, seconds=6, seed=9)
    cfg = {"audio_rate": 16000, "video_rate": 25, "seq_length": 4.8, "feat_size": 0.2}
    sticks = D.StickDataset(folder, normalize="minmax")

    def run(make, epochs=2):
        torch.manual_seed(123)
        out = []
        loader = make()          # (make_loaders seeds numpy itself: the crop draws follow its split, as in the script)
        for _ in range(epochs):
            out.extend(loader)
        return out, torch.rand(1), np.random.rand()

    def same(a, b):
        (ba, ta, na), (bb, tb, nb) = a, b
        assert len(ba) == len(bb) and len(ba) > 2 and torch.equal(ta, tb) and na == nb
        for x, y in zip(ba, bb):
            x, y = (x, y) if isinstance(x, (tuple, list)) else ((x,), (y,))
            assert len(x) == len(y)
            for u, v in zip(x, y):
                if torch.is_tensor(u):
                    assert u.dtype == v.dtype and torch.equal(u, v)
                else:
                    assert list(u) == list(v)

    # phase 3: poses + audio, WeightedRandomSampler, last batch partial
    def p3(device):
        ds = D.SequenceDataset(folder, cfg, scaler=sticks.scaler, withaudio=True)
        ds.truncate()
        return D.make_loaders(ds, 4, withaudio=True, device=device)[0]
    same(run(lambda: p3(None)), run(lambda: p3("cpu")))
    assert isinstance(p3("cpu"), D.ResidentLoader) and len(p3("cpu")) == len(p3(None))

    # phase 2: poses only, SubsetRandomSampler, drop_last
    def p2(resident):
        np.random.seed(5)
        ds = D.SequenceDataset(folder, cfg, scaler=sticks.scaler, withaudio=False)
        sampler = SubsetRandomSampler(range(9))
        if resident:
            return D.ResidentLoader(ds, 4, sampler, "cpu", drop_last=True)
        return DataLoader(ds, batch_size=4, drop_last=True, sampler=sampler, collate_fn=lambda b: D.collate_fn(b, withaudio=False))
    same(run(lambda: p2(False)), run(lambda: p2(True)))

    # phase 1: still poses
    def p1(resident):
        np.random.seed(5)
        sampler = SubsetRandomSampler(range(min(700, len(sticks))))
        if resident:
            return D.ResidentLoader(sticks, 64, sampler, "cpu", drop_last=True)
        return DataLoader(sticks, batch_size=64, drop_last=True, sampler=sampler)
    same(run(lambda: p1(False)), run(lambda: p1(True)))
