"""Dataset side of the train scripts: the callers and data formats just before the hot path
(SURVEY.md 8(f) row 4). Same names, arguments and results as the reference's utils.py:15-194 and the
split / sampler block of phase3/train.py:114-162, so that `phase*/train*.py` run on a
*Music-to-Dance-Motion-Synthesis* folder as well as on `--synthetic` batches.

What differs from the reference, on purpose:
  * MinMax scaling is a small class of our own (`MinMaxScaler`, the subset of sklearn's the reference uses:
    fit / transform / fit_transform / inverse_transform with `data_min_`, `data_max_`, `scale_`, `min_`), with a
    device form (`transform_device` / `inverse_transform_device`: one HIP kernel, m2d_affine_cols) for tensors
    that already live in HBM - the sampling paths of phase2/train.py:192-193 and phase3/test.py:92-101;
  * wav files are read with scipy.io.wavfile (librosa is not a dependency): 16-bit / 32-bit PCM is scaled to
    [-1, 1) and multi-channel audio averaged, which is what `librosa.load(path, sr=None)` returns for them;
  * the takes of a dataset are stored columnar (one array + offsets, `_Ragged`); `collate_fn(..., device=)` pads with
    one masked gather, on the device when asked (default: host tensors, as the reference returns them), and
    `SequenceDataset.sample_batch(indices, device=)` cuts a whole batch of windows out of the HBM-resident store;
  * `SequenceDataset(crop_rng=)`: an explicit generator for the window starts (default: numpy's global one, as in the
    reference); `make_loaders` returns no validation loader when the hold-out is empty.
Nothing here is on the timed path; the kernels it calls are declared in include/m2d.h.
"""
import json
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from . import kernels


# --------------------------------------------------------------------------------------- MinMax scaling
class MinMaxScaler:
    """sklearn.preprocessing.MinMaxScaler(feature_range=(0, 1)) as the reference uses it (utils.py:26-31,79-85):
    per-feature `scale_ = 1 / (max - min)` (1 where max == min), `min_ = -min * scale_`; transform X * scale_ + min_."""

    def __init__(self):
        self.data_min_ = self.data_max_ = self.data_range_ = self.scale_ = self.min_ = None
        self._dev = {}

    def fit(self, X):
        X = np.asarray(X, dtype=np.float64)
        self.data_min_ = X.min(axis=0)
        self.data_max_ = X.max(axis=0)
        self.data_range_ = self.data_max_ - self.data_min_
        rng = self.data_range_.copy()
        rng[rng < 10 * np.finfo(rng.dtype).eps] = 1.0  # sklearn's _handle_zeros_in_scale: constant features keep scale 1
        self.scale_ = 1.0 / rng
        self.min_ = 0.0 - self.data_min_ * self.scale_
        self.n_features_in_ = X.shape[1]
        self._dev = {}
        return self

    def transform(self, X):
        X = np.array(X, dtype=np.float64, copy=True)
        X *= self.scale_
        X += self.min_
        return X

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, X):
        X = np.array(X, dtype=np.float64, copy=True)
        X -= self.min_
        X /= self.scale_
        return X

    # device forms: (..., n_features) fp32 tensors in HBM, one kernel each
    def _coeffs(self, device, inverse):
        key = (str(device), inverse)
        c = self._dev.get(key)
        if c is None:
            if inverse:
                a, b = 1.0 / self.scale_, -self.min_ / self.scale_
            else:
                a, b = self.scale_, self.min_
            c = self._dev[key] = (torch.as_tensor(a, dtype=torch.float32).to(device),
                                  torch.as_tensor(b, dtype=torch.float32).to(device))
        return c

    def transform_device(self, x, out=None):
        a, b = self._coeffs(x.device, False)
        return kernels.impl().affine_cols(x.contiguous(), a, b, out)

    def inverse_transform_device(self, x, out=None):
        a, b = self._coeffs(x.device, True)
        return kernels.impl().affine_cols(x.contiguous(), a, b, out)


# --------------------------------------------------------------------------------------- file loaders
def _read_wav(path):
    """librosa.load(path, sr=None) for PCM / float wav files: float32 mono in [-1, 1)."""
    from scipy.io import wavfile
    _, data = wavfile.read(path)
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    return data


def _dance_dirs(name):
    for directory in os.listdir("{}".format(name)):
        directory = "{}/{}".format(name, directory)
        base = os.path.basename(directory)
        if os.path.isdir(directory) and base[0:5] == "DANCE" and base[-3:] != "bis":
            yield directory, base


def _skeleton_file(base):
    return "/new_skeletons.json" if base[6] == "W" else "/skeletons.json"


def load_sticks(name):
    """utils.py:147-162: the skeleton JSON of every DANCE_* folder (waltz folders use new_skeletons.json)."""
    sticks = []
    for directory, base in _dance_dirs(name):
        file = _skeleton_file(base)
        if os.path.exists(directory + file):
            with open(directory + file) as f:
                sticks.append(json.load(f))
    return sticks


def load_all(name, dance_types, augment=False):
    """utils.py:165-194: skeleton JSONs, resampled audio tracks, style letters and folder names."""
    sticks, musics, labels, dirs = [], [], [], []
    for directory, base in _dance_dirs(name):
        if base[6] not in dance_types:
            continue
        dirs.append(directory)
        labels.append(base[6])
        musics.append(_read_wav(directory + "/resampled_audio_extract.wav"))
        with open(directory + _skeleton_file(base)) as f:
            sticks.append(json.load(f))
    return sticks, musics, labels, dirs


def stickwise(dataset, attribute):
    """utils.py:196-201: all frames of all sequences, concatenated ('skeletons' | 'center')."""
    return np.concatenate([np.asarray(seq[attribute]) for seq in dataset])


def one_hot_encode(labels):
    """utils.py:320-326 (an index encoding, despite the name): position of the style letter in 'CRTW'."""
    dance_types = "CRTW"
    return np.asarray([dance_types.index(l) for l in labels]).astype(int)


def get_positions(sequence, length=120):
    """utils.py:245-248: a random crop [s, s + length) drawn from numpy's global generator."""
    s = np.random.randint(0, len(sequence) - length)
    return s, s + length


# --------------------------------------------------------------------------------------- datasets
# Storage is columnar: every take of a dataset lives in ONE array, takes back to back, with an offsets table - poses
# (sum of frames, 23, 3) and audio (sum of samples,). `sequences` / `musics` hand out per-take views of those arrays
# (the attribute contract of utils.py:48-125), a crop is an index range into them, and a whole batch of crops is one
# gather - on the host, or on the device once `to_device()` has put the two arrays into HBM (a few hundred MB for the
# real dataset against 288 GB: the dataset is resident, the loader moves indices).
class _Ragged:
    """Takes of different lengths stored back to back. Indexing gives a view; assigning a take re-packs."""

    def __init__(self, takes, dtype=None):
        takes = [np.asarray(t) if dtype is None else np.asarray(t, dtype=dtype) for t in takes]
        self._pack(takes)

    def _pack(self, takes):
        self.starts = np.zeros(len(takes) + 1, dtype=np.int64)
        if takes:
            np.cumsum([len(t) for t in takes], out=self.starts[1:])
            self.flat = np.concatenate(takes, axis=0)
        else:
            self.flat = np.zeros((0,))
        self._dev = None

    def __len__(self):
        return len(self.starts) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        return self.flat[self.starts[i]:self.starts[i + 1]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __setitem__(self, i, take):
        takes = list(self)
        takes[int(i)] = np.asarray(take)
        self._pack(takes)

    def lengths(self):
        return np.diff(self.starts)

    def map_rows(self, fn):
        """flat <- fn(flat) on all takes at once (row count unchanged)."""
        out = fn(self.flat)
        assert len(out) == len(self.flat)
        self.flat, self._dev = out, None

    def keep_heads(self, counts):
        """Truncate take i to its first counts[i] rows."""
        counts = np.minimum(np.asarray(counts, dtype=np.int64), self.lengths())
        rows = np.concatenate([np.arange(s, s + c) for s, c in zip(self.starts[:-1], counts)]) if len(self) else []
        self.flat = self.flat[rows]
        self.starts = np.concatenate([[0], np.cumsum(counts)])
        self._dev = None

    def device(self, device, dtype=torch.float32):
        if self._dev is None or self._dev.device != torch.device(device):
            self._dev = torch.from_numpy(np.ascontiguousarray(self.flat)).to(device=device, dtype=dtype)
        return self._dev


def _flatten_features(a):
    return a.reshape(len(a), -1)


class StickDataset(Dataset):
    """All still poses of the dataset, (N, 23, 3); optionally MinMax-scaled to [0, 1] - phase 1's samples, and the
    scaler every phase shares. Interface of utils.py:15-45: `name` = dataset folder, or with resume=True an .npy
    file / array of poses; centering=False adds the per-frame body centre back; attributes skeletons / centers / scaler."""

    def __init__(self, name, resume=False, centering=True, normalize=None):
        if resume:
            poses = np.load(name) if isinstance(name, (str, bytes, os.PathLike)) else np.asarray(name)
        else:
            takes = load_sticks(name)
            poses = stickwise(takes, "skeletons")
            self.centers = stickwise(takes, "center")
            if not centering:
                poses = poses + self.centers[:, None, :]
        self.scaler = None
        if normalize == "minmax":
            self.scaler = MinMaxScaler().fit(_flatten_features(poses))
            poses = self.scaler.transform(_flatten_features(poses)).reshape(poses.shape)
        self.skeletons = poses

    static_items = True  # an item is a fixed row (ResidentLoader gathers a whole epoch at once)

    def __len__(self):
        return self.skeletons.shape[0]

    def __getitem__(self, idx):
        return torch.as_tensor(self.skeletons[idx], dtype=torch.float32)

    def sample_batch(self, indices, device=None):
        """default_collate([self[i] for i in indices]) as ONE row gather - from the HBM-resident copy with `device`."""
        where = device if device is not None else "cpu"
        if getattr(self, "_dev", None) is None or self._dev.device != torch.device(where):
            self._dev = torch.as_tensor(self.skeletons, dtype=torch.float32).to(where)
        return self._dev[torch.as_tensor([int(i) for i in indices], device=where)]

    def statistics(self):
        return self.skeletons.mean(axis=0), self.skeletons.std(axis=0)

    def export(self, path):
        np.save(path, self.skeletons)


class SequenceDataset(Dataset):
    """Pose takes (+ their audio tracks); an item is a random window of `seq_length` seconds (utils.py:48-125).
    `name`: the dataset folder, or with resume=True a dict {'sequences', 'labels', 'dirs'[, 'musics']}.
    `crop_rng` (extension): a numpy RandomState / Generator the window starts are drawn from; default = numpy's global
    generator, which is what the reference draws from (utils.py:245-248), so seeded runs crop identically."""

    def __init__(self, name, config, resume=False, scaler=None, dance_types=("W", "C", "R", "T"), withaudio=False,
                 crop_rng=None):
        self.aud_rate, self.vid_rate = config["audio_rate"], config["video_rate"]
        self.seq_length, self.feat_size = config["seq_length"], config["feat_size"]
        self.stick_length = int(self.seq_length * self.vid_rate)
        self.audio_length = int(self.seq_length * self.aud_rate)
        self.ratio = int(self.aud_rate / self.vid_rate)
        self.withaudio = withaudio
        self.crop_rng = crop_rng
        self.scaler = scaler
        if resume:
            poses, self.labels, self.dirs = name["sequences"], name["labels"], name["dirs"]
            tracks = name["musics"] if withaudio else None
        else:
            takes, tracks, letters, self.dirs = load_all(name, list(dance_types))
            self.labels = one_hot_encode(letters)
            poses = [t["skeletons"] for t in takes]
        self._poses = _Ragged(poses)
        self._tracks = None if tracks is None else _Ragged(tracks)
        if scaler is not None:  # one transform over every frame of every take
            self._poses.map_rows(lambda f: scaler.transform(_flatten_features(f)).reshape(f.shape))

    # the reference's attribute surface: per-take arrays (views of the columnar store)
    @property
    def sequences(self):
        return self._poses

    @property
    def musics(self):
        if self._tracks is None:
            raise AttributeError("musics")
        return self._tracks

    @musics.setter
    def musics(self, tracks):
        self._tracks = tracks if isinstance(tracks, _Ragged) else _Ragged(tracks)

    def __len__(self):
        return len(self._poses)

    def _draw_start(self, idx):
        span = int(self._poses.starts[idx + 1] - self._poses.starts[idx]) - self.stick_length
        rng = self.crop_rng if self.crop_rng is not None else np.random
        draw = getattr(rng, "integers", None) or rng.randint
        return int(draw(0, span))

    def __getitem__(self, idx):
        idx = int(idx)
        first = self._draw_start(idx)
        item = [torch.from_numpy(self._poses[idx][first:first + self.stick_length])]
        if self.withaudio:
            a0 = first * self.ratio
            item.append(torch.from_numpy(self._tracks[idx][a0:a0 + self.audio_length]).float())
        return (*item, torch.as_tensor(np.asarray(self.labels[idx])), self.dirs[idx])

    def sample_batch(self, indices, device=None):
        """A whole batch of windows in one gather - from the HBM-resident store when `device` is given. Same draws, in
        the order of `indices`, and same result as collate_fn([self[i] for i in indices], device=device) (all windows
        have equal length, so the collate step's sort is the identity): (poses, lengths, [audio,] labels, dirs)."""
        indices = [int(i) for i in indices]
        firsts = np.asarray([self._draw_start(i) for i in indices], dtype=np.int64)
        rows = self._poses.starts[indices] + firsts
        where = device if device is not None else "cpu"
        t = torch.arange(self.stick_length, device=where)
        poses = self._poses.device(where)[torch.as_tensor(rows, device=where)[:, None] + t]
        labels = torch.as_tensor(np.asarray([self.labels[i] for i in indices]), device=where)
        dirs = tuple(self.dirs[i] for i in indices)
        lengths = [self.stick_length] * len(indices)
        if not self.withaudio:
            return poses, lengths, labels, dirs
        cols = self._tracks.starts[indices] + firsts * self.ratio
        s = torch.arange(self.audio_length, device=where)
        audio = self._tracks.device(where)[torch.as_tensor(cols, device=where)[:, None] + s]
        return poses, lengths, audio, labels, dirs

    def resample_audio(self, new_rate):
        from scipy.signal import resample
        factor = new_rate / self.aud_rate
        self._tracks = _Ragged([resample(t, int(len(t) * factor)) for t in self._tracks])
        self.aud_rate, self.ratio = new_rate, int(new_rate / self.vid_rate)
        self.audio_length = int(self.seq_length * new_rate)

    def truncate(self):
        """Poses and audio of every take cut to the whole seconds both have (utils.py:109-116)."""
        seconds = np.minimum((self._tracks.lengths() / self.aud_rate).astype(np.int64),
                             (self._poses.lengths() / self.vid_rate).astype(np.int64))
        self._tracks.keep_heads((seconds * self.aud_rate).astype(np.int64))
        self._poses.keep_heads((seconds * self.vid_rate).astype(np.int64))

    def subset(self, indices, withaudio=None):
        """A dataset over the takes `indices` (same rates / crop source)."""
        withaudio = self.withaudio if withaudio is None else withaudio
        state = {"sequences": [self._poses[i] for i in indices], "labels": [self.labels[i] for i in indices],
                 "dirs": [self.dirs[i] for i in indices]}
        if withaudio:
            state["musics"] = [self._tracks[i] for i in indices]
        cfg = {"audio_rate": self.aud_rate, "video_rate": self.vid_rate, "seq_length": self.seq_length,
               "feat_size": self.feat_size}
        return SequenceDataset(state, cfg, resume=True, withaudio=withaudio, crop_rng=self.crop_rng)

    def export(self, pathfile):
        with open(pathfile, "wb") as f:
            pickle.dump({"sequences": list(self._poses), "labels": self.labels, "dirs": self.dirs}, f)


def collate_fn(batch, withaudio=True, device=None):
    """Ragged samples -> one zero-padded batch, longest sample first (what pack_padded_sequence wants; utils.py:128-144):
    (poses (B, Tmax, 23, 3) fp32, lengths, [audio (B, S),] labels (B,), dirs). The padding is ONE masked gather over
    the concatenated frames instead of a copy per sample; with `device` the frames are moved first and gather, mask
    and the other tensors are made there."""
    order = sorted(range(len(batch)), key=lambda i: -len(batch[i][0]))  # stable: ties keep their order, like list.sort
    cols = list(zip(*(batch[i] for i in order)))
    frames, dirs, labels = cols[0], cols[-1], torch.stack(cols[-2])
    lengths = [len(f) for f in frames]
    where = device if device is not None else frames[0].device
    flat = torch.cat([f.reshape(len(f), -1) for f in frames]).to(device=where, dtype=torch.float32, non_blocking=True)
    n = torch.as_tensor(lengths, device=where)
    t = torch.arange(max(lengths), device=where)
    inside = t[None, :] < n[:, None]                                      # (B, Tmax)
    src = (torch.cumsum(n, 0) - n)[:, None] + torch.where(inside, t[None, :], torch.zeros_like(t)[None, :])
    padded = (flat[src] * inside[..., None]).reshape(len(frames), len(t), *frames[0].shape[1:])
    labels = labels.to(where, non_blocking=True)
    if not withaudio:
        return padded, lengths, labels, dirs
    return padded, lengths, torch.stack(cols[1]).to(where, non_blocking=True), labels, dirs


class ResidentLoader:
    """`DataLoader(dataset, batch_size, sampler=sampler, drop_last=..., collate_fn=collate_fn)` for a dataset that lives
    in HBM: the train scripts' loaders (phase1/train_wgan-gp.py:71-72, phase2/train.py:115-116, phase3/train.py:150-162)
    fetch 64 items and collate them on the host, 5-33 ms per batch against 0.8-12 ms of GPU work per loop body; here the
    whole dataset (the 61 takes of the reference's data are < 1 GB of fp32) is uploaded once and a batch is the sampler's
    indices + the window draws on the host and ONE gather on the device (`dataset.sample_batch(indices, device)`).
    Draw for draw the DataLoader's stream: its iterator's base-seed draw from torch's global generator, the sampler's
    draws, the datasets' crop draws in item order - a seeded run sees the same batches either way
    (tests/test_data_pipeline.py). `sampler` is any torch sampler over item indices."""

    def __init__(self, dataset, batch_size, sampler, device, drop_last=False):
        self.dataset, self.sampler, self.device = dataset, sampler, device
        self.batch_size, self.drop_last = int(batch_size), bool(drop_last)

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()  # (what a DataLoader iterator draws for its workers' base seed)
        if getattr(self.dataset, "static_items", False):
            # items without per-draw randomness (still poses): the whole epoch in ONE gather, batches are views of it
            order = list(self.sampler)
            rows = self.dataset.sample_batch(order, self.device) if order else None
            stop = len(order) - (len(order) % self.batch_size if self.drop_last else 0)
            for i in range(0, stop, self.batch_size):
                yield rows[i:i + self.batch_size]
            return
        chunk = []
        for i in self.sampler:
            chunk.append(i)
            if len(chunk) == self.batch_size:
                yield self.dataset.sample_batch(chunk, self.device)
                chunk = []
        if chunk and not self.drop_last:
            yield self.dataset.sample_batch(chunk, self.device)


# --------------------------------------------------------------------------------------- split + samplers
def split_indices(dataset_size, validation_split=.2, test_split=.5, random_seed=14):
    """The seeded hold-out of phase3/train.py:112-124 -> (train, val, test) index lists: a permutation of range(n)
    under numpy seed 14; its first floor(.2 n) entries are held out, the first floor(half) of those for test. Seeds
    numpy's GLOBAL generator like the reference does (the crops drawn afterwards depend on that state)."""
    np.random.seed(random_seed)
    perm = np.random.permutation(dataset_size).tolist()
    held = int(validation_split * dataset_size)
    test = int(test_split * held)
    return perm[held:], perm[test:held], perm[:test]


def class_balanced_weights(labels, indices):
    """phase3/train.py:132-143: per-sample weight = 1 / (number of samples of its style among `indices`)."""
    labels = np.asarray(labels)
    # (the reference indexes 1/counts with the label itself, which needs every style 0..k-1 present in the subset;
    # indexing through the inverse map gives the same weights then and still works when a style is missing)
    _, inverse, counts = np.unique(labels[indices], return_inverse=True, return_counts=True)
    return (1. / counts)[inverse]


def make_loaders(dataset, batch_size, withaudio=True, logdir=None, device=None):
    """Train / validation loaders over the seeded split (phase3/train.py:112-162): trainvaltest_samples.json in
    `logdir`, each subset drawn through a class-balanced WeightedRandomSampler, validation served as ONE batch.
    A dataset too small to hold anything out (< 5 takes) gets no validation loader (None). `device`: the loaders keep
    their subsets in that device's memory and gather batches there (ResidentLoader; same batches as the DataLoader)."""
    from torch.utils.data import DataLoader, WeightedRandomSampler
    parts = dict(zip(("train", "val", "test"), split_indices(len(dataset))))
    if logdir is not None:
        with open(os.path.join(logdir, "trainvaltest_samples.json"), "w") as f:
            json.dump({k + "_samples": [dataset.dirs[i] for i in v] for k, v in parts.items()}, f)

    def loader(idx, per_batch):
        if not idx:
            return None
        weights = class_balanced_weights(dataset.labels, idx)
        sampler = WeightedRandomSampler(weights, len(weights))
        if device is not None:
            return ResidentLoader(dataset.subset(idx, withaudio), per_batch, sampler, device)
        return DataLoader(dataset.subset(idx, withaudio), batch_size=per_batch, sampler=sampler,
                          collate_fn=lambda b: collate_fn(b, withaudio=withaudio))

    return loader(parts["train"], batch_size), loader(parts["val"], len(parts["val"])), tuple(parts.values())


# --------------------------------------------------------------------------------------- synthetic dataset folder
def write_synthetic_dataset(folder, n_takes=8, seconds=8, audio_rate=16000, video_rate=25, seed=0,
                            styles="CRTW"):
    """A folder in the on-disk format the loaders above read (DANCE_<style>_<n>/{skeletons.json | new_skeletons.json,
    config.json, resampled_audio_extract.wav}) filled with random poses / audio: lets the train scripts' dataset
    path run end to end without the real (absent) Music-to-Dance-Motion-Synthesis data."""
    from scipy.io import wavfile
    rng = np.random.RandomState(seed)
    os.makedirs(folder, exist_ok=True)
    for n in range(n_takes):
        style = styles[n % len(styles)]
        d = os.path.join(folder, "DANCE_%s_%d" % (style, n + 1))
        os.makedirs(d, exist_ok=True)
        frames = seconds * video_rate + int(rng.randint(0, video_rate))
        walk = np.cumsum(rng.randn(frames, 23, 3) * 0.02, axis=0) + rng.randn(1, 23, 3)
        center = np.cumsum(rng.randn(frames, 3) * 0.01, axis=0)
        with open(d + _skeleton_file("DANCE_%s" % style), "w") as f:
            json.dump({"skeletons": walk.tolist(), "center": center.tolist()}, f)
        with open(d + "/config.json", "w") as f:
            json.dump({"start_position": 0, "end_position": frames}, f)
        samples = seconds * audio_rate + int(rng.randint(0, audio_rate))
        pcm = np.clip(rng.randn(samples) * 3000.0, -32767, 32767).astype(np.int16)
        wavfile.write(d + "/resampled_audio_extract.wav", audio_rate, pcm)
    return folder
