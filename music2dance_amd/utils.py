"""Host-side helpers of the hot path: weight initialisation and audio windowing.

`initialize_weights` restates utils.py:267-313 of the reference; `slice_audio_batch`
replaces the reference's O(T^2) torch.cat loop (utils.py:329-353) by a strided view of the
padded track, bit-identical to it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def initialize_weights(net, initialisation=None, bias=None):
    """Xavier-normal (or N(mean, std) when `initialisation=(mean, std)`) on every
    Conv / ConvTranspose / Linear weight and every GRU `weight_*` matrix, in
    `net.modules()` order; biases keep their constructor values unless `bias` is given,
    in which case conv / linear biases are zeroed. BatchNorm is untouched."""
    dense = (nn.Conv1d, nn.Conv2d, nn.Linear, nn.ConvTranspose1d, nn.ConvTranspose2d)
    for m in net.modules():
        if isinstance(m, dense):
            if initialisation is None:
                nn.init.xavier_normal_(m.weight)
            else:
                m.weight.data.normal_(initialisation[0], initialisation[1])
            if bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.GRU):
            for names in m._all_weights:
                for name in names:
                    if "weight" not in name:
                        continue
                    if initialisation is None:
                        nn.init.xavier_normal_(m._parameters[name])
                    else:
                        nn.init.normal_(m._parameters[name], initialisation[0], initialisation[1])


def slice_audio_sequence(seq, audio_feat_samples, cutting_stride, pad_samples, device=None):
    """(samples,) -> (n_windows, audio_feat_samples): zero-pad pad_samples//2 on the left and
    the remainder on the right, then one window every `cutting_stride` samples."""
    left = pad_samples // 2
    padded = F.pad(seq, (left, pad_samples - left))
    return padded.unfold(0, audio_feat_samples, cutting_stride).contiguous()


def slice_audio_batch(batch, audio_feat_samples, cutting_stride, pad_samples, device="cpu", lazy=False):
    """utils.slice_audio_batch of the reference on whatever device `batch` lives on:
    (B, samples) -> (B, n_windows, audio_feat_samples) (or the 1-D form).
    lazy=True returns the SAME values as an overlapping-window view of the padded track (no copy:
    1/5 of the bytes at window 3200 / hop 640); the phase-3 generator reads such a view in place
    (its first encoder conv gathers the windows from the track), everything else can call
    .contiguous() on it."""
    if batch.dim() == 1:
        return slice_audio_sequence(batch, audio_feat_samples, cutting_stride, pad_samples)
    left = pad_samples // 2
    padded = F.pad(batch, (left, pad_samples - left))
    view = padded.unfold(-1, audio_feat_samples, cutting_stride)
    return view if lazy else view.contiguous()


def nparams(model):
    return sum(p.numel() for p in model.parameters())


# ---- sampling helpers (utils.py:205-242 of the reference): eval-mode generation ----------
def sampleG(model, noise=None, device="cpu"):
    """Phase-1 generator sample(s): one (23, 3) pose when `noise` is None, else model(noise) as numpy."""
    model.eval()
    with torch.no_grad():
        if noise is None:
            z = torch.randn(1, model.latent_size, device=device)
            return model(z)[0].detach().cpu().numpy().reshape(23, 3)
        return model(noise).detach().cpu().numpy()


def sampleseqG(model, stick_length, noise=None, device="cpu"):
    """Phase-2 sequence sample(s) of `stick_length` frames -> (frames[, x batch], 23, 3) numpy."""
    model.eval()
    with torch.no_grad():
        if noise is None:
            z = torch.randn(1, stick_length, model.input_size, device=device)
            return model(z, [stick_length]).detach().cpu().numpy().reshape(stick_length, 23, 3)
        out = model(noise, [stick_length] * noise.shape[0]).detach().cpu().numpy()
        return out.reshape(stick_length * noise.shape[0], 23, 3)


def sampleaudioG(model, audio_slices, noise=None):
    """Phase-3 sample: poses for a (B, T, window) batch of audio windows of ANY length T (the
    generator is length-agnostic; the reference samples 750-frame videos, phase3/test.py:49)."""
    model.eval()
    with torch.no_grad():
        B, T = audio_slices.shape[0], audio_slices.shape[1]
        out = model(audio_slices, [T] * B, noise)
        return out.detach().cpu().numpy().reshape(B * T, 23, 3)
