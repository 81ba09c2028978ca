"""torchsparse.nn modules (core/models/build_blocks.py:25-80).

State-dict compatibility with the reference checkpoints: the conv weight is the
parameter ``kernel`` of shape [K, Cin, Cout] ([Cin, Cout] when K == 1), init
U(+-1/sqrt(fan * K)) with fan = Cout if transposed else Cin (v1.4.0).
BatchNorm / ReLU stay nn.BatchNorm1d / nn.ReLU subclasses so that
SparseSyncBatchNorm.convert_sync_batchnorm (core/models/utils.py:179) finds them.
"""
import math

import numpy as np
import torch
from torch import nn

from ..utils import make_ntuple
from . import functional as F
from .utils import fapply

__all__ = ['Conv3d', 'BatchNorm', 'ReLU', 'LeakyReLU']


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1,
                 bias=False, transposed=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = make_ntuple(kernel_size, ndim=3)
        self.stride = make_ntuple(stride, ndim=3)
        self.dilation = dilation
        self.transposed = transposed
        self.kernel_volume = int(np.prod(self.kernel_size))
        if self.kernel_volume > 1:
            self.kernel = nn.Parameter(torch.zeros(self.kernel_volume, in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.zeros(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def extra_repr(self):
        s = '{in_channels}, {out_channels}, kernel_size={kernel_size}'
        if self.stride != (1,) * len(self.stride):
            s += ', stride={stride}'
        if self.dilation != 1:
            s += ', dilation={dilation}'
        if self.bias is None:
            s += ', bias=False'
        if self.transposed:
            s += ', transposed=True'
        return s.format(**self.__dict__)

    def reset_parameters(self):
        std = 1 / math.sqrt((self.out_channels if self.transposed else self.in_channels)
                            * self.kernel_volume)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def forward(self, input):
        return F.conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias,
                        stride=self.stride, dilation=self.dilation, transposed=self.transposed)


class BatchNorm(nn.BatchNorm1d):
    """nn.BatchNorm1d over SparseTensor.feats on the HIP BatchNorm kernels."""

    def forward(self, input):
        return fapply(input, F.batch_norm, self)


class ReLU(nn.ReLU):
    def forward(self, input):
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input):
        return fapply(input, super().forward)
