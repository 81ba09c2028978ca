"""Sparse conv blocks with the reference's module layout
(core/models/build_blocks.py:21-83) so state-dict keys match:
``net.0.kernel``, ``net.1.{weight,bias,running_*}``, ``downsample.0.kernel`` ..."""
from torch import nn

from ..torchsparse import nn as spnn

__all__ = ['BasicConvolutionBlock', 'BasicDeconvolutionBlock', 'ResidualBlock']


def _conv_bn(inc, outc, ks, stride=1, dilation=1, transposed=False, relu=True):
    layers = [spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, transposed=transposed),
              spnn.BatchNorm(outc)]
    if relu:
        layers.append(spnn.ReLU(True))
    return layers


class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(*_conv_bn(inc, outc, ks, stride, dilation))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(*_conv_bn(inc, outc, ks, stride, transposed=True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(*_conv_bn(inc, outc, ks, stride, dilation),
                                 *_conv_bn(outc, outc, ks, 1, dilation, relu=False))
        if inc == outc and stride == 1:
            self.downsample = nn.Sequential()
        else:
            self.downsample = nn.Sequential(*_conv_bn(inc, outc, 1, stride, 1, relu=False))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        return self.relu(self.net(x) + self.downsample(x))
