"""Sparse conv blocks with the reference's module layout
(core/models/build_blocks.py:21-83) so state-dict keys match:
``net.0.kernel``, ``net.1.{weight,bias,running_*}``, ``downsample.0.kernel`` ..."""
import torch
from torch import nn

from ..torchsparse import SparseTensor
from ..torchsparse import nn as spnn
from ..torchsparse.nn import functional as spf
from ..torchsparse.nn.utils import fapply

__all__ = ['BasicConvolutionBlock', 'BasicDeconvolutionBlock', 'ResidualBlock', 'FusedSequential',
           'PointBatchNorm1d', 'PointLinear']


class PointLinear(nn.Linear):
    """nn.Linear over point features [N, C] on the HIP MFMA pipeline (the reference's
    point_transforms use plain nn.Linear, spvcnn.py:58-74; same parameters / state-dict keys).
    rocBLAS picks a 32x32 macro-tile kernel for these tall-skinny fp32 products (~10 TFLOP/s at
    80k x 256 x 128); the pair kernel's dense mode runs them at the conv's rate."""

    def forward(self, input):
        return spf.linear(input, self.weight, self.bias)


class PointBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d over point features [N, C] on the HIP BatchNorm kernels (the
    reference's point_transforms use plain nn.BatchNorm1d, spvcnn.py:58-74; same keys)."""

    def forward(self, input):
        return spf.batch_norm(input, self)


# U2MKD_FOLD_EVAL_BN=0: inference runs spnn.Conv3d and its eval-mode BatchNorm as two passes (the formulation the folded one is
# tested against)
import os as _os
_FOLD_EVAL_BN = _os.environ.get('U2MKD_FOLD_EVAL_BN', '1') != '0'
_FOLDABLE_BN = ('BatchNorm', 'SparseSyncBatchNorm')      # (eval mode: a SyncBatchNorm uses its running statistics too)

# BatchNorm flavours whose forward is spf.batch_norm (plain and the SyncBatchNorm conversions of point_voxel.py)
_FUSABLE_BN = ('BatchNorm', 'PointBatchNorm1d', 'SparseSyncBatchNorm', 'PointSyncBatchNorm1d')
_STATS_BN = ('BatchNorm',)      # the BatchNorm whose statistics pass a convolution's store can stand in for (spnn.BatchNorm, not synchronised)


class FusedSequential(nn.Sequential):
    """nn.Sequential with the reference's layout (same state-dict keys) whose forward
    runs every BatchNorm -> ReLU pair as ONE fused HIP pass (BN statistics, normalise,
    affine and max(.,0) in a single read/write of the feature matrix)."""

    def forward(self, x, residual=None):
        """``residual`` (a SparseTensor / tensor like the output): the sequence must END in a BatchNorm, and the
        result is relu(sequence(x) + residual) with the add and the ReLU inside that BatchNorm's pass."""
        mods = list(self)
        i = 0
        # inference: spnn.Conv3d -> eval-mode BatchNorm (-> ReLU | + residual -> ReLU) as ONE convolution whose store applies the
        # folded affine (functional.conv_eval_affine): the frozen teacher's ~49 BatchNorm passes disappear
        fold = _FOLD_EVAL_BN and isinstance(x, SparseTensor) and not torch.is_grad_enabled()
        stats = None
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if fold and isinstance(m, spnn.Conv3d) and type(nxt).__name__ in _FOLDABLE_BN and not nxt.training \
                    and nxt.track_running_stats and nxt.running_mean is not None:
                after = mods[i + 2] if i + 2 < len(mods) else None
                last_bn = residual is not None and i + 1 == len(mods) - 1
                relu = last_bn or type(after) in (spnn.ReLU, nn.ReLU)
                scale, shift = spf.eval_bn_affine(nxt)
                y = spf.conv_eval_affine(x, m, scale, shift, relu, residual if last_bn else None)
                if y is not None:
                    x = y
                    i += 2 if last_bn or not relu else 3
                    continue
            if (isinstance(m, spnn.Conv3d) and m.bias is None and type(nxt).__name__ in _STATS_BN and nxt.training
                    and isinstance(x, SparseTensor) and torch.is_grad_enabled() and spf.conv_bn_stats_enabled()):
                # spnn.Conv3d -> train-mode BatchNorm: the convolution's store may take the BatchNorm's slab statistics with it
                # (functional.BnStats: the wide layers' gather-sum does), the BatchNorm then starts at its merge step
                stats = spf.BnStats()
                spf.BN_STATS_SINK[0] = stats
                try:
                    x = m(x)
                finally:
                    spf.BN_STATS_SINK[0] = None
                i += 1
                continue
            st, stats = (stats if stats is not None and stats.partial is not None else None), None      # (what the previous module's store left)
            fusable = (type(m).__name__ in _FUSABLE_BN and type(nxt) in (spnn.ReLU, nn.ReLU))
            if residual is not None and i == len(mods) - 1:
                assert type(m).__name__ in _FUSABLE_BN, type(m).__name__
                r = residual.F if isinstance(residual, SparseTensor) else residual
                kw = {'stats': st} if st is not None else {}      # (the call's shape stays batch_norm(x, bn, relu, residual) otherwise)
                if isinstance(x, SparseTensor):
                    x = fapply(x, spf.batch_norm, m, True, r, **kw)
                else:
                    x = spf.batch_norm(x, m, True, r, **kw)
                i += 1
            elif fusable:
                kw = {'stats': st} if st is not None else {}
                if isinstance(x, SparseTensor):
                    x = fapply(x, spf.batch_norm, m, True, **kw)
                else:
                    x = spf.batch_norm(x, m, True, **kw)
                i += 2
            elif st is not None and st.partial is not None and type(m).__name__ in _STATS_BN:
                x = fapply(x, spf.batch_norm, m, False, None, stats=st)      # (BatchNorm without a ReLU behind it)
                i += 1
            elif isinstance(m, PointLinear) and type(nxt).__name__ in _FUSABLE_BN and nxt.training and not isinstance(x, SparseTensor):
                x = spf.linear(x, m.weight, m.bias, bias_feeds_batchnorm=True)     # (its bias gradient is identically zero)
                i += 1
            else:
                x = m(x)
                i += 1
        return x


def _conv_bn(inc, outc, ks, stride=1, dilation=1, transposed=False, relu=True):
    layers = [spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, transposed=transposed),
              spnn.BatchNorm(outc)]
    if relu:
        layers.append(spnn.ReLU(True))
    return layers


class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = FusedSequential(*_conv_bn(inc, outc, ks, stride, dilation))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = FusedSequential(*_conv_bn(inc, outc, ks, stride, transposed=True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = FusedSequential(*_conv_bn(inc, outc, ks, stride, dilation),
                                   *_conv_bn(outc, outc, ks, 1, dilation, relu=False))
        if inc == outc and stride == 1:
            self.downsample = nn.Sequential()
        else:
            self.downsample = FusedSequential(*_conv_bn(inc, outc, 1, stride, 1, relu=False))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        # relu(net(x) + downsample(x)): the add and the ReLU run inside the last BatchNorm's pass (forward and backward)
        return self.net(x, residual=self.downsample(x))
