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


def slice_audio_batch(batch, audio_feat_samples, cutting_stride, pad_samples, device="cpu"):
    """utils.slice_audio_batch of the reference on whatever device `batch` lives on:
    (B, samples) -> (B, n_windows, audio_feat_samples) (or the 1-D form)."""
    if batch.dim() == 1:
        return slice_audio_sequence(batch, audio_feat_samples, cutting_stride, pad_samples)
    left = pad_samples // 2
    padded = F.pad(batch, (left, pad_samples - left))
    return padded.unfold(-1, audio_feat_samples, cutting_stride).contiguous()


def nparams(model):
    return sum(p.numel() for p in model.parameters())
