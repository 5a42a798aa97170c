"""Deterministic weights and inputs shared by the fixture generator (reference side), the
oracle tests and the GPU parity tests. Inputs are never stored: every side regenerates
them from these seeded formulas, and the fixtures carry float64 checksums so that any
drift of the generator (torch version, platform) is detected before outputs are compared.
"""
import math

import torch

WINDOW, HOP, PAD = 3200, 640, 2560  # phase3/train.py:80-83 with 16 kHz audio, 25 fps


def _gen(seed):
    return torch.Generator().manual_seed(int(seed))


def fill_state_dict(sd, base_seed=1000):
    """Returns a new dict with every float tensor of `sd` replaced, in key order, by a
    seeded pattern: Xavier-scaled normal for matrices / conv kernels, small normal for
    biases, 1 + 0.1 N for BatchNorm gammas, [0.7, 1.3] for running variances."""
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        if not torch.is_floating_point(v):
            out[k] = torch.zeros_like(v)  # num_batches_tracked restarts at 0
            continue
        g = _gen(base_seed + i)
        if k.endswith("running_var"):
            t = 1.0 + 0.3 * (2 * torch.rand(v.shape, generator=g) - 1)
        elif k.endswith("running_mean"):
            t = 0.05 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1 and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:
            t = 0.05 * torch.randn(v.shape, generator=g)
        else:
            rf = 1
            for s in v.shape[2:]:
                rf *= s
            std = math.sqrt(2.0 / (v.shape[1] * rf + v.shape[0] * rf))
            t = std * torch.randn(v.shape, generator=g)
        out[k] = t.to(v.dtype)
    return out


def poses(B, T, seed=11):
    """MinMax-scaled poses live in [0, 1) (utils.py:26-31): (B, T, 69)."""
    return torch.rand(B, T, 69, generator=_gen(seed))


def audio(B, T, seed=12):
    """(B, 640*T) samples ~ N(0, 0.1^2)."""
    return 0.1 * torch.randn(B, HOP * T, generator=_gen(seed))


def noise(B, T, size, seed=13):
    return torch.randn(B, T, size, generator=_gen(seed))


def alpha(B, seed=14):
    return torch.rand(B, 1, generator=_gen(seed))


def slices(aud):
    """utils.slice_audio_batch(audio, 3200, 640, 2560) (verified bit-identical)."""
    return torch.nn.functional.pad(aud, (PAD // 2, PAD - PAD // 2)).unfold(-1, WINDOW, HOP).contiguous()


def checksum(t):
    t = t.detach().double()
    return [t.sum().item(), t.abs().sum().item()]


def sd_checksums(sd):
    import numpy as np
    return np.array([checksum(v.float() if not torch.is_floating_point(v) else v) for v in sd.values()])


def template(keys, shapes):
    """Zero state_dict with the reference's key names / shapes (stored in the fixtures)."""
    sd = {}
    for k, sh in zip(keys, shapes):
        k, sh = str(k), str(sh)
        shape = tuple(int(d) for d in sh.split(",")) if sh else ()
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shape, dtype=torch.int64)
        else:
            sd[k] = torch.zeros(shape, dtype=torch.float32)
    return sd
