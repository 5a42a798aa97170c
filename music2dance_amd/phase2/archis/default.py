"""Phase-2 unconditional sequence GAN (GRU generator, temporal-conv critic) on HIP kernels.

Constructor signatures, attribute names and state_dict keys follow the reference
(phase2/archis/default.py:5-49,90-163).
"""
import torch
import torch.nn as nn

from ... import ops
from ...layers import GRU, BatchNorm1d, Conv1d, Linear, batched_bn_counters, lengths_tensor
from ...utils import initialize_weights


def _check_sorted(lengths):
    ls = [int(v) for v in lengths]
    if any(a < b for a, b in zip(ls, ls[1:])):
        # same failure mode as pack_padded_sequence(enforce_sorted=True) in the reference
        raise RuntimeError("`lengths` array must be sorted in decreasing order")
    return ls


class NoiseGen(nn.Module):
    """nn.GRU wrapper returning the output sequence only (phase2/archis/default.py:90-96)."""

    def __init__(self, input_size, output_size, n_layers):
        super().__init__()
        self.rnn = GRU(input_size, output_size, n_layers, batch_first=True)

    def forward(self, x, lengths=None):
        return self.rnn(x, lengths)[0]


class LinearBlock(nn.Module):
    """x + relu(bn2(fc2(x))) with the reference's dead fc1 -> bn1 branch kept only for its
    running-statistics side effect (phase2/archis/default.py:136-145)."""

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
    """Per-frame residual MLP: latent -> size -> nblocks x LinearBlock -> pose (69)."""

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
    """Two 'same' temporal convolutions with ReLU and a skip connection."""

    def __init__(self, channels, ksize):
        super().__init__()
        self.channels, self.ksize = channels, ksize
        self.pad = int((ksize - 1) / 2)
        self.conv1 = Conv1d(channels, channels, kernel_size=ksize, padding=self.pad, dilation=1)
        self.conv2 = Conv1d(channels, channels, kernel_size=ksize, padding=self.pad, dilation=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        # conv1 -> conv2 is a sole-consumer edge: pre-masked gradients (ops.conv1d, in_act / out_pm)
        h = self.conv2(self.conv1(x, act=ops.ACT_RELU, out_pm=True), act=ops.ACT_RELU,
                       in_act=(ops.ACT_RELU, 0.0))
        return x + h


class SequenceGenerator(nn.Module):
    def __init__(self, input_size, latent_size, size, output_size, n_blocks, n_cells=1, device="cpu"):
        super().__init__()
        self.input_size, self.latent_size, self.size, self.output_size = input_size, latent_size, size, output_size
        self.noise_gen = NoiseGen(input_size, latent_size, n_cells)
        self.decoder = FrameDecoder(latent_size, size, output_size, n_blocks)
        initialize_weights(self)
        self.to(device)

    def forward(self, x, lengths):
        ls = _check_sorted(lengths)
        with batched_bn_counters(self):  # one fused add for the decoder's BatchNorm step counters
            h = self.noise_gen(x, lengths_tensor(ls, x.size(1), x.device))
            h = h[:, :max(ls)]
            return self.decoder(h.reshape(-1, self.decoder.latent_size))


class SequenceDiscriminator(nn.Module):
    """TCN critic: conv(k=init_ker) + ReLU, n_blocks TemporalBlocks, full-length conv -> score."""

    def __init__(self, channels_in, channels_h, seqlen, init_ker=7, n_blocks=1, device="cpu"):
        super().__init__()
        self.conv1 = Conv1d(channels_in, channels_h, kernel_size=init_ker, padding=int((init_ker - 1) / 2))
        self.blocks = nn.Sequential(*[TemporalBlock(channels_h, 7) for _ in range(n_blocks)])
        self.lastconv = Conv1d(channels_h, 1, seqlen)
        self.relu = nn.ReLU(inplace=True)
        initialize_weights(self)
        self.to(device)

    def forward(self, x):
        h = self.blocks(self.conv1(x, act=ops.ACT_RELU))
        return self.lastconv(h).squeeze(1)

    def score_pair(self, x_a, x_b):
        """critic(x_a), critic(x_b) from one pass over the concatenated batch."""
        n = x_a.size(0)
        s = self.forward(torch.cat((x_a, x_b), 0))
        return s[:n], s[n:]
