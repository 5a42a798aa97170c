"""WGAN-GP / WGAN-LP gradient penalty and total-variation loss on HIP kernels.

Same signatures as the reference's losses.py:5-60,76-82. The penalty is the composition
    interpolate (fused lerp kernel) -> critic forward -> autograd.grad w.r.t. the inputs
    (first backward, recorded with create_graph=True through the closed conv / linear op
    set of ops.py) -> per-sample L2 norm + penalty (fused reduction kernel)
and stays differentiable w.r.t. the critic's parameters.
"""
import torch
import torch.autograd as autograd

from . import ops
from .layers import host_draw, to_device_async  # noqa: F401


def gradient_penalty(critic, bsize, real, fake, audio=None, is_seq=False, is_cond=False, lp=False, device=None,
                     alpha=None):
    """Gradient penalty for the stick (phase 1) and sequence (phase 2/3) WGAN frameworks.

    lp=False: WGAN-GP  mean_b (sqrt(sum g_b^2 + 1e-12) - 1)^2
    lp=True : WGAN-LP  mean_b max(0, ||g_b||_2 - 1)^2   (no eps)
    With `audio` the critic takes (poses, audio) and the penalty is the SUM of the pose term
    and the audio term. As in the reference, alpha ~ U(0,1) per sample comes from the
    default HOST generator (losses.py:15) and `audio.requires_grad_(True)` is applied to
    the caller's tensor (losses.py:26-27). `alpha` (extension, (bsize, 1) on the device): use
    these interpolation weights instead of drawing them (captured-graph replays feed the draw
    through a static buffer)."""
    real2d = real.reshape(real.size(0), -1)
    fake2d = fake.reshape(fake.size(0), -1)
    if alpha is None:
        alpha = host_draw("rand", (bsize, 1), real2d.device)
    interpol = ops.gp_interpolate(real2d, fake2d, alpha.view(-1))
    interpol = interpol.view(interpol.size(0), 69, -1) if is_seq else interpol.view(interpol.size(0), 23, 3)
    interpol.requires_grad_(True)
    if audio is not None:
        audio.requires_grad_(True)
        score = critic(interpol, audio)
        if is_cond:
            score = score[0]
        inputs = (interpol, audio)
    else:
        score = critic(interpol)
        if is_cond:
            score = score[0]
        inputs = (interpol,)
    with ops.input_grads_only():
        grads = autograd.grad(outputs=score, inputs=inputs, grad_outputs=torch.ones_like(score),
                              create_graph=True, retain_graph=True, only_inputs=True)
    g0 = grads[0].reshape(grads[0].size(0), -1)
    if audio is None:
        return ops.gp_penalty(g0, lp=lp)
    g1 = grads[1].reshape(grads[1].size(0), -1)
    return ops.gp_penalty(g0, lp=False) + ops.gp_penalty(g1, lp=False)


def tv_loss(sequence):
    """Total-variation regulariser: mean |x[:, :, 1:] - x[:, :, :-1]| of a (B, C, T) tensor."""
    return ops.tv_mean(sequence)


def jerkiness(sequence):
    """Evaluation metric of the reference (losses.py:85-89, phase3/test.py:78-104): the squared third finite
    difference along time of a (B, C, T) sequence, summed over channels, averaged over (B, T - 3)."""
    return ops.jerk_mean(sequence)
