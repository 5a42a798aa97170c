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
  * `collate_fn(..., device=)` can place the padded batch on the device directly (default: host tensors, as the
    reference returns them).
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
class StickDataset(Dataset):
    """Still poses (phase 1) and the fit of the MinMax scaler every phase shares (utils.py:15-45)."""

    def __init__(self, name, resume=False, centering=True, normalize=None):
        self.scaler = None
        if resume:
            self.skeletons = np.load(name) if isinstance(name, (str, bytes, os.PathLike)) else np.asarray(name)
        else:
            sticks = load_sticks(name)
            self.skeletons = stickwise(sticks, "skeletons")
            self.centers = stickwise(sticks, "center")
            if not centering:
                self.skeletons = self.skeletons + self.centers[:, np.newaxis]
        if normalize == "minmax":
            self.scaler = MinMaxScaler()
            dshape = np.shape(self.skeletons)
            flat = np.reshape(self.skeletons, (dshape[0], -1))
            self.skeletons = np.reshape(self.scaler.fit_transform(flat), dshape)

    def __len__(self):
        return len(self.skeletons)

    def __getitem__(self, idx):
        return torch.from_numpy(self.skeletons[idx]).float()

    def statistics(self):
        return self.skeletons.mean(0), self.skeletons.std(0)

    def export(self, path):
        np.save(path, self.skeletons)


class SequenceDataset(Dataset):
    """Pose sequences (+ audio tracks) with random fixed-length crops (utils.py:48-125). `name` is the dataset
    folder, or with resume=True a dict {'sequences', 'labels', 'dirs'[, 'musics']}."""

    def __init__(self, name, config, resume=False, scaler=None, dance_types=["W", "C", "R", "T"], withaudio=False):
        self.scaler = None
        self.aud_rate = config["audio_rate"]
        self.vid_rate = config["video_rate"]
        self.seq_length = config["seq_length"]
        self.stick_length = int(config["seq_length"] * self.vid_rate)
        self.audio_length = int(config["seq_length"] * self.aud_rate)
        self.ratio = int(config["audio_rate"] / config["video_rate"])
        self.feat_size = config["feat_size"]
        self.withaudio = withaudio
        if resume:
            self.sequences = name["sequences"]
            self.labels = name["labels"]
            self.dirs = name["dirs"]
            if withaudio:
                self.musics = name["musics"]
        else:
            sticks, musics, labels, dirs = load_all(name, dance_types)
            self.labels = one_hot_encode(labels)
            self.dirs = dirs
            self.musics = musics
            self.sequences = [np.asarray(s["skeletons"]) for s in sticks]
        if scaler is not None:
            self.scaler = scaler
            for i, seq in enumerate(self.sequences):
                dshape = np.shape(seq)
                flat = self.scaler.transform(np.reshape(seq, (seq.shape[0], -1)))
                self.sequences[i] = np.reshape(flat, dshape)

    def __len__(self):
        return len(self.sequences)

    def __getitem__(self, idx):
        s, e = get_positions(self.sequences[idx], length=self.stick_length)
        label = torch.from_numpy(np.asarray(self.labels[idx]))
        if not self.withaudio:
            return torch.from_numpy(self.sequences[idx][s:e]), label, self.dirs[idx]
        s_a = s * self.ratio
        e_a = s_a + self.audio_length
        return (torch.from_numpy(self.sequences[idx][s:e]), torch.from_numpy(self.musics[idx][s_a:e_a]).float(),
                label, self.dirs[idx])

    def resample_audio(self, new_rate):
        from scipy.signal import resample
        for i in range(len(self.musics)):
            self.musics[i] = resample(self.musics[i], int(len(self.musics[i]) * new_rate / self.aud_rate))
        self.aud_rate = new_rate
        self.ratio = int(new_rate / self.vid_rate)
        self.audio_length = int(self.seq_length * new_rate)

    def truncate(self):
        """Cut audio and poses of every take to their common whole number of seconds (utils.py:109-116)."""
        for i in range(len(self)):
            al = int(len(self.musics[i]) / self.aud_rate)
            sl = int(len(self.sequences[i]) / self.vid_rate)
            mini = min(al, sl)
            self.musics[i] = self.musics[i][:int(mini * self.aud_rate)]
            self.sequences[i] = self.sequences[i][:int(mini * self.vid_rate)]

    def export(self, pathfile):
        with open(pathfile, "wb") as f:
            pickle.dump({"sequences": self.sequences, "labels": self.labels, "dirs": self.dirs}, f)


def collate_fn(batch, withaudio=True, device=None):
    """utils.py:128-144: sort the samples by length (longest first), zero-pad the poses to (B, Tmax, 23, 3) fp32.
    -> (padded_seqs, lengths, [musics,] labels, dirs). device: build / move the tensors there (extension)."""
    batch.sort(key=lambda x: len(x[0]), reverse=True)
    if withaudio:
        sequences, musics, labels, dirs = zip(*batch)
        musics = torch.stack(musics)
    else:
        sequences, labels, dirs = zip(*batch)
    labels = torch.stack(labels)
    lengths = [len(seq) for seq in sequences]
    padded_seqs = torch.zeros(len(sequences), max(lengths), 23, 3)
    for i, seq in enumerate(sequences):
        padded_seqs[i, :lengths[i]] = seq[:lengths[i]]
    if device is not None:
        padded_seqs, labels = padded_seqs.to(device, non_blocking=True), labels.to(device, non_blocking=True)
        if withaudio:
            musics = musics.to(device, non_blocking=True)
    if withaudio:
        return padded_seqs, lengths, musics, labels, dirs
    return padded_seqs, lengths, labels, dirs


# --------------------------------------------------------------------------------------- split + samplers
def split_indices(dataset_size, validation_split=.2, test_split=.5, random_seed=14):
    """phase3/train.py:112-124: shuffle range(n) with numpy seed 14; the first floor(.2 n) indices are held out,
    half of them (floor) for test. -> (train, val, test) index lists."""
    indices = list(range(dataset_size))
    vsplit = int(np.floor(validation_split * dataset_size))
    tsplit = int(np.floor(test_split * vsplit))
    np.random.seed(random_seed)
    np.random.shuffle(indices)
    return indices[vsplit:], indices[tsplit:vsplit], indices[:tsplit]


def class_balanced_weights(labels, indices):
    """phase3/train.py:132-143: per-sample weight = 1 / (number of samples of its style among `indices`)."""
    labels = np.asarray(labels)
    # (the reference indexes 1/counts with the label itself, which needs every style 0..k-1 present in the subset;
    # indexing through the inverse map gives the same weights then and still works when a style is missing)
    _, inverse, counts = np.unique(labels[indices], return_inverse=True, return_counts=True)
    return (1. / counts)[inverse]


def make_loaders(dataset, batch_size, withaudio=True, logdir=None):
    """The train / validation loaders of phase3/train.py:112-162: seeded split, trainvaltest_samples.json,
    class-balanced WeightedRandomSamplers, resume-datasets of the two subsets, validation served as one batch."""
    from torch.utils.data import DataLoader, WeightedRandomSampler
    train_idx, val_idx, test_idx = split_indices(len(dataset))
    if logdir is not None:
        with open(logdir + "/trainvaltest_samples.json", "w+") as f:
            json.dump({"train_samples": [dataset.dirs[i] for i in train_idx],
                       "val_samples": [dataset.dirs[i] for i in val_idx],
                       "test_samples": [dataset.dirs[i] for i in test_idx]}, f)
    cfg = {"audio_rate": dataset.aud_rate, "video_rate": dataset.vid_rate, "seq_length": dataset.seq_length,
           "feat_size": dataset.feat_size}
    loaders = []
    for idx, bs in ((train_idx, batch_size), (val_idx, len(val_idx))):
        w = class_balanced_weights(dataset.labels, idx)
        sub = {"sequences": [dataset.sequences[i] for i in idx], "labels": [dataset.labels[i] for i in idx],
               "dirs": [dataset.dirs[i] for i in idx]}
        if withaudio:
            sub["musics"] = [dataset.musics[i] for i in idx]
        ds = SequenceDataset(sub, cfg, resume=True, withaudio=withaudio)
        loaders.append(DataLoader(ds, batch_size=max(bs, 1), sampler=WeightedRandomSampler(w, len(w)),
                                  collate_fn=lambda b, wa=withaudio: collate_fn(b, withaudio=wa)))
    return loaders[0], loaders[1], (train_idx, val_idx, test_idx)


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
