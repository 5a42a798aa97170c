"""Phase-1 still-pose residual-MLP generator / critic on the HIP kernels.

Same constructors, attributes and state_dict keys as the reference
(phase1/archis/residual.py:4-71); forward passes run as fused GEMM(+bias+ReLU) and
BatchNorm(+ReLU+residual) launches.
"""
import torch
import torch.nn as nn

from ... import ops
from ...layers import BatchNorm1d, Dropout, Linear


class LinearBlock(nn.Module):
    """x + relu(bn2(fc2(x))). The reference also evaluates fc1 -> bn1 -> relu and then
    overwrites it (phase1/archis/residual.py:63-71): fc1 / bn1 never influence the output or
    receive gradients, but bn1's running statistics advance in training mode — reproduced
    here by a gradient-free observation pass."""

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


class Generator(nn.Module):
    def __init__(self, latent_size, size, output_size, nblocks):
        super().__init__()
        self.latent_size, self.size, self.output_size, self.nblocks = latent_size, size, output_size, nblocks
        self.fc1 = Linear(latent_size, size)
        self.bn1 = BatchNorm1d(size, eps=1e-5, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.blocks = nn.Sequential(*[LinearBlock(size, use_bn=True) for _ in range(nblocks)])
        self.dropout = Dropout(p=0.5)
        self.lastfc = Linear(size, output_size)

    def forward(self, x):
        h = self.bn1(self.fc1(x), act=ops.ACT_RELU)
        h = self.blocks(h)
        return self.lastfc(self.dropout(h))


class Discriminator(nn.Module):
    def __init__(self, input_size, size, nblocks):
        super().__init__()
        self.input_size, self.size, self.nblocks = input_size, size, nblocks
        self.fc1 = Linear(input_size, size)
        self.relu = nn.ReLU(inplace=True)
        self.blocks = nn.Sequential(*[LinearBlock(size) for _ in range(nblocks)])
        self.dropout = Dropout(p=0.5)
        self.lastfc = Linear(size, 1)

    def forward(self, x):
        h = self.fc1(x.reshape(x.size(0), -1), act=ops.ACT_RELU)
        h = self.blocks(h)
        return self.lastfc(self.dropout(h))
