"""Phase-3 audio-conditioned generator and critics on the gfx950 kernels.

Constructor signatures, attribute names and state_dict keys follow the reference
(phase3/archis/default.py:6-355) so checkpoints and train scripts are interchangeable;
forward passes launch fused conv(+bias+ReLU), BatchNorm(+ReLU / LeakyReLU), GEMM and GRU
step kernels.

One addition: `SequenceDiscriminator.shared_audio()`. The audio branch of the critic sees
the same, never-interpolated audio in the gradient-penalty, real and fake passes of an
iteration (losses.py:29, phase3/train.py:204-211), so inside that context its code is
computed once and shared (identical scores).
"""
import contextlib
import os

import torch
import torch.nn as nn

from ... import kernels, ops
from ...layers import (GRU, BatchNorm1d, Conv1d, Linear, WindowView, batched_bn_counters, head_activation,
                       lengths_tensor, to_device_async)
from ...utils import initialize_weights


def _descending(lengths):
    ls = [int(v) for v in lengths]
    if any(a < b for a, b in zip(ls, ls[1:])):
        raise RuntimeError("`lengths` array must be sorted in decreasing order")
    return ls


_FUSED_BN_STATS = os.environ.get("M2D_FUSED_BN_STATS", "1") != "0"  # dev switch for A/B timing
RELU_IN = (ops.ACT_RELU, 0.0)  # "my input is the sole-consumer output of a fused conv + ReLU" (ops.conv1d)


_FUSED_BN_THEN = os.environ.get("M2D_FUSED_BN_THEN", "1") != "0"  # A/B lever: the U-Net's pool / upsample inside the BatchNorm pass


def _conv_bn(conv, bn, x, act, slope=0.0, into=None, then=None, then_out=None):
    """bn(conv(x)) + activation. In training mode the conv's epilogue hands the BatchNorm its batch
    statistics (no second pass over the activation); x may be a WindowView of the padded track
    (first encoder conv: the audio windows are read in place). into: ops.batch_norm's `out`.
    then (no autograd graph only): the pass that follows in the U-Net - "pool": -> (y, max_pool(y)); "upsample": ->
    upsample2_linear(y) written into `then_out` - made inside the normalisation pass when the batch statistics come from the
    conv's epilogue (layers.BatchNorm1d.forward_then), by the separate kernels otherwise."""
    stats = bn.training and _FUSED_BN_STATS
    if isinstance(x, WindowView):
        out = conv.forward_windows(x.track, x.T, x.hop, x.window, with_stats=stats)
    else:
        out = conv(x, with_stats=stats)
    y, sums = out if stats else (out, None)
    if _BN_SUMS_LOG[0] is not None:
        # (SequenceGenerator.audio_path(keep=True): the batch statistics every BatchNorm of the audio path was advanced
        # with - None when this one computed its own: the path then cannot be replayed)
        _BN_SUMS_LOG[0].append((bn, sums if (stats and not ops.sync_batchnorm_active()) else None, y.numel() // y.shape[1]))
    if then is None:
        return bn(y, act=act, slope=slope, sums=sums, out=into)
    assert not torch.is_grad_enabled()
    if _FUSED_BN_THEN and stats and not ops.sync_batchnorm_active() and y.dim() == 3:
        L = y.shape[2]
        if then == "pool" and L % 2 == 0:
            return bn.forward_then(y, sums, "pool", act, slope, out=into)
        if then == "upsample" and (y.shape[1] * L) % 2 == 0:
            return bn.forward_then(y, sums, "upsample", act, slope, then_out=then_out)
    z = bn(y, act=act, slope=slope, sums=sums, out=into)
    if then == "pool":
        return z, ops.maxpool2(z)
    return ops.upsample2_linear(z, out=then_out)


_BN_SUMS_LOG = [None]


def _head(module, conv, x, in_act=None):
    """last conv of an encoder / critic branch followed by the 'id'|'relu'|'tanh' switch"""
    y = conv(x, act=module._head_act, in_act=in_act)
    return ops.tanh(y) if module._head_tanh else y


class NoiseGen(nn.Module):
    def __init__(self, input_size, output_size, n_layers):
        super().__init__()
        self.rnn = GRU(input_size, output_size, n_layers, batch_first=True)

    def forward(self, x, lengths=None):
        return self.rnn(x, lengths)[0]


class LinearBlock(nn.Module):
    """x + relu(bn2(fc2(x))); fc1 / bn1 are the reference's dead branch
    (phase3/archis/default.py:183-192): only bn1's running statistics observe it."""

    def __init__(self, size, use_bn=False):
        super().__init__()
        self.size = size
        self.use_bn = use_bn
        self.fc1 = Linear(size, size, bias=True)
        self.fc2 = Linear(size, size, bias=True)
        if use_bn:
            self.bn1 = BatchNorm1d(size, eps=1e-5, momentum=0.1)
            self.bn2 = BatchNorm1d(size, eps=1e-5, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if not self.use_bn:
            return x + self.fc2(x, act=ops.ACT_RELU)
        if self.training:
            with torch.no_grad():
                self.bn1.observe(self.fc1(x.detach()))
        return self.bn2(self.fc2(x), act=ops.ACT_RELU, residual=x)


class FrameDecoder(nn.Module):
    def __init__(self, latent_size, size, output_size, nblocks):
        super().__init__()
        self.latent_size, self.size, self.output_size, self.nblocks = latent_size, size, output_size, nblocks
        self.fc1 = Linear(latent_size, size)
        self.bn1 = BatchNorm1d(size, eps=1e-5, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.blocks = nn.Sequential(*[LinearBlock(size, use_bn=True) for _ in range(nblocks)])
        self.lastfc = Linear(size, output_size)

    def forward(self, x):
        h = self.bn1(self.fc1(x), act=ops.ACT_RELU)
        return self.lastfc(self.blocks(h))


class TemporalBlock(nn.Module):
    def __init__(self, channels, ksize):
        super().__init__()
        self.channels, self.ksize = channels, ksize
        self.pad = int((ksize - 1) / 2)
        self.conv1 = Conv1d(channels, channels, kernel_size=ksize, padding=self.pad, dilation=1)
        self.conv2 = Conv1d(channels, channels, kernel_size=ksize, padding=self.pad, dilation=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        # conv1's output feeds conv2 only: conv2's backward hands conv1 a gradient already multiplied
        # by relu'(conv1 output); conv2's own output also feeds the residual add, so it keeps its mask
        h = self.conv2(self.conv1(x, act=ops.ACT_RELU, out_pm=True), act=ops.ACT_RELU, in_act=RELU_IN)
        return x + h


# --------------------------------------------------------------------------------------- encoders
class DefaultAudioEncoder(nn.Module):
    """3200-sample window -> code: conv(k250,s50) then five stride-2 convs, each followed by
    BatchNorm + ReLU, and a k=2 head (phase3/archis/default.py:59-82)."""

    def __init__(self, f_maps, output_size, activ="id"):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        self.activations = nn.ModuleList()
        self.conv_layers.append(Conv1d(1, f_maps, 250, 50, 124))
        self.activations.append(nn.Sequential(BatchNorm1d(f_maps), nn.ReLU(True)))
        for _ in range(5):
            self.conv_layers.append(Conv1d(f_maps, f_maps * 2, 4, 2, 1))
            self.activations.append(nn.Sequential(BatchNorm1d(f_maps * 2), nn.ReLU(True)))
            f_maps *= 2
        self.conv_layers.append(Conv1d(f_maps, output_size, 2))
        mod, self._head_act, self._head_tanh = head_activation(activ)
        self.activations.append(mod)

    def forward(self, x):
        for conv, post in zip(self.conv_layers[:-1], self.activations[:-1]):
            x = _conv_bn(conv, post[0], x, ops.ACT_RELU)
        return _head(self, self.conv_layers[-1], x).squeeze()


class BasisConvBlock(nn.Module):
    def __init__(self, channels_in, channels_out):
        super().__init__()
        self.conv = Conv1d(channels_in, channels_out, 3, 1, 1)
        self.bn = BatchNorm1d(channels_out)
        self.relu = nn.LeakyReLU(0.2)

    def forward(self, x, out=None, then=None, then_out=None):
        return _conv_bn(self.conv, self.bn, x, ops.ACT_LEAKY, 0.2, into=out, then=then, then_out=then_out)


class UBlock(nn.Module):
    """Four-level 1-D U-Net: conv blocks, MaxPool(2,2) down, linear x2 upsampling + skip concat up."""

    def __init__(self, channels):
        super().__init__()
        self.convblock1 = BasisConvBlock(channels, channels)
        self.convblock2 = BasisConvBlock(channels, channels)
        self.convblock3 = BasisConvBlock(channels, channels)
        self.convblock4 = BasisConvBlock(channels, channels)
        self.convblock5 = BasisConvBlock(channels * 2, channels)
        self.convblock6 = BasisConvBlock(channels * 2, channels)
        self.convblock7 = BasisConvBlock(channels * 2, channels)
        self.downsample = nn.MaxPool1d(2, 2)
        self.upsample = nn.Upsample(scale_factor=2, mode="linear", align_corners=False)

    def forward(self, x):
        if not torch.is_grad_enabled() and x.dim() == 3 and x.shape[2] % 8 == 0:
            # no autograd graph (the critic iterations' generator forward, validation, sampling): the three skip
            # concatenations are never copied together - each skip's BatchNorm and the upsampling write the two channel
            # halves of one (B, 2C, L) buffer (m2d_bn_fwd_sums_to / m2d_upsample2_fwd_to). Same values, same order.
            B, C, L = x.shape
            cat1 = torch.empty((B, 2 * C, L), dtype=x.dtype, device=x.device)
            cat2 = torch.empty((B, 2 * C, L // 2), dtype=x.dtype, device=x.device)
            cat3 = torch.empty((B, 2 * C, L // 4), dtype=x.dtype, device=x.device)
            # (round 6: a skip's max-pool and a decoder level's upsampling are made inside the BatchNorm pass that
            # produces their input - `then`: the normalised decoder tensors d4 / u3 / u2 are never written)
            _, p1 = self.convblock1(x, out=cat1[:, C:], then="pool")
            _, p2 = self.convblock2(p1, out=cat2[:, C:], then="pool")
            _, p3 = self.convblock3(p2, out=cat3[:, C:], then="pool")
            self.convblock4(p3, then="upsample", then_out=cat3[:, :C])
            self.convblock5(cat3, then="upsample", then_out=cat2[:, :C])
            self.convblock6(cat2, then="upsample", then_out=cat1[:, :C])
            return self.convblock7(cat1)
        d1 = self.convblock1(x)
        d2 = self.convblock2(ops.maxpool2(d1))
        d3 = self.convblock3(ops.maxpool2(d2))
        d4 = self.convblock4(ops.maxpool2(d3))
        u3 = self.convblock5(torch.cat((ops.upsample2_linear(d4), d3), 1))
        u2 = self.convblock6(torch.cat((ops.upsample2_linear(u3), d2), 1))
        return self.convblock7(torch.cat((ops.upsample2_linear(u2), d1), 1))


class UNetAudioEncoder(nn.Module):
    def __init__(self, f_maps, output_size, activ="id"):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        self.activations = nn.ModuleList()
        self.conv_layers.append(Conv1d(1, f_maps, 160, 4, 79))
        self.activations.append(nn.Sequential(BatchNorm1d(f_maps), nn.LeakyReLU(0.2)))
        for _ in range(2):
            self.conv_layers.append(Conv1d(f_maps, f_maps * 2, 4, 2, 1))
            self.activations.append(nn.Sequential(BatchNorm1d(f_maps * 2), nn.LeakyReLU(0.2)))
            f_maps *= 2
        self.ublock = UBlock(f_maps)
        self.fc = Conv1d(f_maps, output_size, 200)
        self.activ, self._head_act, self._head_tanh = head_activation(activ)

    def forward(self, x):
        for conv, post in zip(self.conv_layers, self.activations):
            x = _conv_bn(conv, post[0], x, ops.ACT_LEAKY, 0.2)
        return _head(self, self.fc, self.ublock(x)).squeeze()


class WaveGANAudioEncoder(nn.Module):
    def __init__(self, f_maps, output_size, activ="id"):
        super().__init__()
        self.l1 = Conv1d(1, f_maps, 25, stride=4)
        self.bn1 = BatchNorm1d(f_maps)
        self.l2 = Conv1d(f_maps, f_maps * 2, 25, stride=4)
        f_maps *= 2
        self.bn2 = BatchNorm1d(f_maps)
        self.l3 = Conv1d(f_maps, f_maps * 2, 25, stride=4)
        f_maps *= 2
        self.bn3 = BatchNorm1d(f_maps)
        self.l4 = Conv1d(f_maps, f_maps * 2, 25, stride=4)
        f_maps *= 2
        self.bn4 = BatchNorm1d(f_maps)
        self.l5 = Conv1d(f_maps, output_size, 5)
        self.relu = nn.ReLU(True)
        self.activ, self._head_act, self._head_tanh = head_activation(activ)

    def forward(self, x):
        for conv, bn in ((self.l1, self.bn1), (self.l2, self.bn2), (self.l3, self.bn3), (self.l4, self.bn4)):
            x = _conv_bn(conv, bn, x, ops.ACT_RELU)
        return _head(self, self.l5, x).squeeze(-1)


class AudioEncoder(nn.Module):
    def __init__(self, type, f_maps, output_size, activ="id"):
        super().__init__()
        if type == "default":
            self.model = DefaultAudioEncoder(f_maps, output_size, activ)
        elif type == "unet":
            self.model = UNetAudioEncoder(f_maps, output_size, activ)
        elif type == "wavegan":
            self.model = WaveGANAudioEncoder(f_maps, output_size, activ)

    def forward(self, x):
        return self.model(x)


# --------------------------------------------------------------------------------------- generator
class SequenceGenerator(nn.Module):
    """audio windows -> per-frame code (encoder) -> GRU(+ noise GRU) -> per-frame pose decoder."""

    def __init__(self, window_size, input_size, latent_size, size, output_size, noise_size, n_blocks,
                 n_cells=1, enc_type="default", activ="id", device="cpu"):
        super().__init__()
        self.window_size, self.input_size, self.latent_size = window_size, input_size, latent_size
        self.size, self.noise_size, self.output_size = size, noise_size, output_size
        self.device = device
        self.audio_enc = AudioEncoder(enc_type, 32, input_size, activ)
        self.audio_rnn = NoiseGen(input_size, latent_size - noise_size, n_cells)
        self.noise_gen = NoiseGen(noise_size, noise_size, 1)
        self.decoder = FrameDecoder(latent_size, size, output_size, n_blocks)
        initialize_weights(self)
        self.to(device)

    def _side_stream(self, device):
        if device.type != "cuda" or not self.overlap_noise_gru:
            return None
        if getattr(self, "_noise_stream", None) is None:
            self._noise_stream = torch.cuda.Stream(device=device)
        return self._noise_stream

    overlap_noise_gru = True

    def forward(self, x, lengths, noise=None):
        # x: (batch, frames, window)
        with batched_bn_counters(self):  # one fused add for the 12 BatchNorm step counters
            return self._forward(x, lengths, noise)

    def _forward(self, x, lengths, noise=None):
        frames = x.size(1)
        # an overlapping-window view of the padded track (utils.slice_audio_batch(..., lazy=True)) is read
        # in place by the first encoder conv; a dense (B, T, window) tensor is used as the reference does
        wv = WindowView.of(x) if (x.is_cuda and not x.requires_grad and x.size(2) == self.window_size) else None
        enc_in = wv if wv is not None else x.reshape(-1, 1, self.window_size)
        code = self.audio_enc(enc_in).view(-1, frames, self.input_size)
        if noise is None:
            # drawn from the HOST generator, then moved (phase3/archis/default.py:31-34)
            noise = to_device_async(torch.randn(list(code.size()[:-1]) + [self.noise_size]), code.device)
        ls = _descending(lengths)
        # the two recurrences are independent and latency-bound (a few dozen blocks per step):
        # the 1-layer noise GRU runs on a side stream under the 3-layer audio GRU
        side = self._side_stream(code.device)
        if side is not None:
            cur = torch.cuda.current_stream(code.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                n = self.noise_gen(noise)
            # `noise` lives in the caller's (or the staging stream's) allocator pool and is read on the side stream - in the
            # backward pass too, as a saved tensor of the recurrence. Without this the pool may hand its memory out again
            # the moment autograd drops it, while the side stream's weight-gradient GEMM has not read it yet (round 6:
            # tests/test_flip_audit.py found noise_gen.rnn.weight_ih_l0 wrong in 5 runs of 6 under a particular timing)
            noise.record_stream(side)
            h = self.audio_rnn(code, lengths_tensor(ls, frames, code.device))[:, :max(ls)]
            cur.wait_stream(side)
            n.record_stream(cur)
        else:
            h = self.audio_rnn(code, lengths_tensor(ls, frames, code.device))[:, :max(ls)]
            n = self.noise_gen(noise)
        if self._keep_audio_path:
            self._kept_audio = (h, self._kept_audio[1] if self._kept_audio else None)
        latent = torch.cat((h, n), -1)
        return self.decoder(latent.reshape(-1, self.decoder.latent_size))

    # -- the audio path of a forward, kept for a second forward of the SAME batch through the SAME weights --------------
    # The reference's loop body that holds a generator iteration runs the generator twice on one batch: once for the
    # critic iteration (phase3/train.py:195, graph built and dropped) and once for the generator iteration (:222), with
    # fresh noise. Encoder and audio GRU do not see the noise: their activations are the same both times. forward(...,
    # keep_audio_path=True) keeps the audio GRU's output (with its autograd graph) and the batch statistics every
    # BatchNorm of the encoder was advanced with; forward_from_kept_audio_path(noise) is then the second forward:
    # noise GRU + decoder on the kept output, the encoder's running statistics advanced once more with the SAME sums
    # (m2d_bn_update_running: exactly what the second pass would leave), the step counters bumped as forward() does.
    _keep_audio_path = False
    _kept_audio = None

    def forward_keeping_audio_path(self, x, lengths, noise=None):
        """forward(x, lengths, noise) that keeps what forward_from_kept_audio_path needs (call with grad enabled).
        -> rows; self.kept_audio_path() tells whether the second forward may use it."""
        self._kept_audio = None
        log = []
        _BN_SUMS_LOG[0] = log
        self._keep_audio_path = True
        try:
            rows = self.forward(x, lengths, noise)
        finally:
            _BN_SUMS_LOG[0] = None
            self._keep_audio_path = False
        enc_bns = [m for m in self.audio_enc.modules() if isinstance(m, nn.BatchNorm1d)]
        mine = [e for e in log if any(e[0] is m for m in enc_bns)]
        ok = (self._kept_audio is not None and len(mine) == len(enc_bns) and all(e[1] is not None for e in mine)
              and len({id(e[0]) for e in mine}) == len(enc_bns))
        self._kept_audio = (self._kept_audio[0], mine) if ok else None
        return rows

    def kept_audio_path(self):
        return self._kept_audio is not None

    def drop_kept_audio_path(self):
        self._kept_audio = None

    def forward_from_kept_audio_path(self, noise=None, after=None):
        """The second forward of the batch forward_keeping_audio_path saw (same weights in between): -> rows.
        after: the completion event of that first forward when it ran on another stream (the engine's generator stream,
        one loop body ahead): the current stream waits for it and takes the kept tensors over explicitly - they live in
        the other stream's allocator pool (the ordering used to rest on transitive waits)."""
        h, sums_log = self._kept_audio
        self._kept_audio = None
        if h.is_cuda:
            cur = torch.cuda.current_stream(h.device)
            if after is not None:
                cur.wait_event(after)
            h.record_stream(cur)
            for _, sums, _ in sums_log:
                if torch.is_tensor(sums) and sums.is_cuda:
                    sums.record_stream(cur)
        with batched_bn_counters(self):   # every BatchNorm's step counter, the encoder's included: as forward() does
            if self.training:
                with torch.no_grad():
                    for bn, sums, count in sums_log:
                        kernels.impl().bn_update_running(sums, float(count), bn.running_mean, bn.running_var, bn.eps, bn.momentum)
            if noise is None:
                noise = to_device_async(torch.randn(list(h.size()[:-1]) + [self.noise_size]), h.device)
            n = self.noise_gen(noise)
            latent = torch.cat((h, n), -1)
            return self.decoder(latent.reshape(-1, self.decoder.latent_size))


# --------------------------------------------------------------------------------------- critics
class StickDiscriminator(nn.Module):
    """Pose branch: conv(k=init_ker)+ReLU, n_blocks TemporalBlocks, full-length conv -> code."""

    def __init__(self, channels_in, channels_h, output_code, seqlen, init_ker=9, n_blocks=2, activ="id"):
        super().__init__()
        self.conv1 = Conv1d(channels_in, channels_h, kernel_size=init_ker, padding=int((init_ker - 1) / 2))
        self.blocks = nn.Sequential(*[TemporalBlock(channels_h, 7) for _ in range(n_blocks)])
        self.fconv = Conv1d(channels_h, output_code, seqlen)
        self.relu = nn.ReLU(inplace=True)
        self.activ, self._head_act, self._head_tanh = head_activation(activ)

    def forward(self, x):
        h = self.blocks(self.conv1(x, act=ops.ACT_RELU))
        return _head(self, self.fconv, h).squeeze(-1)


class AudioDiscriminator(nn.Module):
    """Raw-audio branch: five k=25 stride-4 convs + ReLU, then a k=75 conv -> code (76 800 samples only)."""

    def __init__(self, output_size, activ="id"):
        super().__init__()
        pad = 11
        self.l1 = Conv1d(1, 32, 25, stride=4, padding=pad)
        self.l2 = Conv1d(32, 64, 25, stride=4, padding=pad)
        self.l3 = Conv1d(64, 128, 25, stride=4, padding=pad)
        self.l4 = Conv1d(128, 256, 25, stride=4, padding=pad)
        self.l5 = Conv1d(256, 512, 25, stride=4, padding=pad)
        self.l6 = Conv1d(512, output_size, 75)
        self.relu = nn.ReLU(True)
        self.activ, self._head_act, self._head_tanh = head_activation(activ)

    def forward(self, x):
        # a pure chain: every layer's output has exactly one consumer, the next conv, so gradients
        # travel pre-multiplied by relu' (epilogue of the consumer's backward-data kernel)
        x = self.l1(x, act=ops.ACT_RELU, out_pm=True)
        for conv in (self.l2, self.l3, self.l4, self.l5):
            x = conv(x, act=ops.ACT_RELU, in_act=RELU_IN, out_pm=True)
        return _head(self, self.l6, x, in_act=RELU_IN).squeeze(-1)


class SequenceDiscriminator(nn.Module):
    def __init__(self, channels_in, channels_h, output_code, seqlen, init_ker=9, activ="id", device="cpu"):
        super().__init__()
        self.stick_d = StickDiscriminator(channels_in, channels_h, output_code, seqlen, init_ker=init_ker,
                                          activ=activ)
        self.audio_d = AudioDiscriminator(output_code, activ)
        self.fc1 = Linear(2 * output_code, 128)
        self.fc2 = Linear(128, 1)
        self.relu = nn.ReLU(True)
        initialize_weights(self)
        self.to(device)
        self._share = None

    @contextlib.contextmanager
    def shared_audio(self):
        """Within the context, repeated calls with the SAME audio tensor object reuse one
        audio_d(audio) evaluation (and one backward through it)."""
        prev = self._share
        self._share = {}
        try:
            yield self
        finally:
            self._share = prev

    def audio_code(self, c):
        """Evaluate (and, inside shared_audio(), cache) the audio branch for `c` ahead of the calls that use it."""
        return self._audio_code(c)

    def _audio_code(self, c):
        if self._share is None:
            return self.audio_d(c)
        key = (id(c), c._version, c.requires_grad)
        hit = self._share.get(key)
        if hit is None:
            hit = (c, self.audio_d(c))  # keep `c` alive so id() stays unique
            self._share[key] = hit
        return hit[1]

    # The pose branch and the audio branch meet only at fc1: the pose branch (small launches that
    # leave most CUs idle between dependent kernels) runs on a side stream under the audio branch's
    # large convolutions. Autograd replays each node's backward on the stream of its forward, so
    # the first, second and final backward passes overlap the same way.
    overlap_branches = os.environ.get("M2D_BRANCH_OVERLAP", "1") != "0"

    def _stick_code(self, x):
        dev = x.device
        if dev.type != "cuda" or not self.overlap_branches:
            return self.stick_d(x)  # (under graph capture the fork / join below is captured as such)
        if getattr(self, "_stick_stream", None) is None:
            self._stick_stream = torch.cuda.Stream(device=dev)
            self._joins = {}
        side, cur = self._stick_stream, torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            code = self.stick_d(x)
        x.record_stream(side)
        self._joins[id(code)] = (side, code)  # holds `code`, so the id stays unique until the join
        return code

    def _joined(self, code):
        j = self._joins.pop(id(code), None) if getattr(self, "_joins", None) else None
        if j is not None:
            cur = torch.cuda.current_stream(code.device)
            cur.wait_stream(j[0])
            code.record_stream(cur)
        return code

    def forward(self, x, c):
        stick = self._stick_code(x)  # enqueued first, on the side stream
        acode = self._audio_code(c)
        code = torch.cat((self._joined(stick), acode), -1)
        return self.fc2(self.fc1(code, act=ops.ACT_RELU))

    def begin_pair(self, x_a, x_b):
        """Start the pose branch of a later score_pair(x_a, x_b, c) now. It needs the two pose batches and the
        branch's weights only, so the training loop issues it BEFORE the gradient penalty: the side stream then
        runs it underneath the penalty pass's audio-branch kernels instead of after them with the main stream idle."""
        self._pair = (x_a, x_b, self._stick_code(torch.cat((x_a, x_b), 0)))

    def score_pair(self, x_a, x_b, c):
        """critic(x_a, c), critic(x_b, c) from ONE pass over the concatenated poses (the critic
        has no cross-sample coupling, so the scores are the per-call ones; twice the columns
        per launch fill the chip better) and one evaluation of the audio branch."""
        n = x_a.size(0)
        pre, self._pair = getattr(self, "_pair", None), None
        if pre is not None and pre[0] is x_a and pre[1] is x_b:
            stick = pre[2]
        else:
            stick = self._stick_code(torch.cat((x_a, x_b), 0))
        acode = self._audio_code(c)
        stick = self._joined(stick)
        code = torch.cat((stick, torch.cat((acode, acode), 0)), -1)
        s = self.fc2(self.fc1(code, act=ops.ACT_RELU))
        return s[:n], s[n:]


class AblatedSequenceDiscriminator(nn.Module):
    """Pose-only critic. Like the reference it does NOT forward `init_ker` to the pose
    branch, which therefore uses its default kernel of 9 (phase3/archis/default.py:277-278)."""

    def __init__(self, channels_in, channels_h, output_code, seqlen, init_ker=9, activ="id", device="cpu"):
        super().__init__()
        self.stick_d = StickDiscriminator(channels_in, channels_h, output_code, seqlen, activ=activ)
        self.fc1 = Linear(output_code, 128)
        self.fc2 = Linear(128, 1)
        self.relu = nn.ReLU(True)
        initialize_weights(self)
        self.to(device)

    def forward(self, x):
        return self.fc2(self.fc1(self.stick_d(x), act=ops.ACT_RELU))

    def score_pair(self, x_a, x_b):
        n = x_a.size(0)
        s = self.forward(torch.cat((x_a, x_b), 0))
        return s[:n], s[n:]
