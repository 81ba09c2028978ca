"""SPVCNN LiDAR-only network (row a9; core/models/semantickitti/spvcnn.py:10-142).

Same sub-module names, channel plan and data flow as the reference, hence the
same state-dict keys and the same per-point logits."""
import torch
from torch import nn

from .. import torchsparse
from ..torchsparse import PointTensor
from ..torchsparse import nn as spnn
from ..torchsparse.nn import functional as spf
from .blocks import (BasicConvolutionBlock, BasicDeconvolutionBlock, FusedSequential, PointBatchNorm1d, PointLinear,
                     ResidualBlock)
from .point_voxel import initial_voxelize, point_to_voxel, prepare_geometry, voxel_to_point

__all__ = ['SPVCNN']

_BASE_CHANNELS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


class SPVCNN(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        cr = kwargs.get('cr')
        cs = [int(cr * c) for c in _BASE_CHANNELS]
        self.cs = cs
        self.in_channel = kwargs.get('in_channel', 4)
        self.num_classes = kwargs.get('num_classes', 17)
        self.out_channel = cs[-1]
        if 'pres' in kwargs and 'vres' in kwargs:
            self.pres = kwargs['pres']
            self.vres = kwargs['vres']

        self.stem = FusedSequential(
            spnn.Conv3d(self.in_channel, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))

        self.vox_downs = nn.ModuleList([
            nn.Sequential(BasicConvolutionBlock(cs[i], cs[i], ks=2, stride=2, dilation=1),
                          ResidualBlock(cs[i], cs[i + 1], ks=3, stride=1, dilation=1),
                          ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))
            for i in range(4)])

        self.vox_ups = nn.ModuleList([
            nn.ModuleList([
                BasicDeconvolutionBlock(cs[i], cs[i + 1], ks=2, stride=2),
                nn.Sequential(
                    ResidualBlock(cs[i + 1] + cs[len(cs) - 2 - i], cs[i + 1], ks=3, stride=1, dilation=1),
                    ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))])
            for i in range(4, len(cs) - 1)])

        self.classifier_vox = nn.Sequential(PointLinear(cs[8], self.num_classes))

        self.point_transforms = nn.ModuleList([
            FusedSequential(PointLinear(cs[a], cs[b]), PointBatchNorm1d(cs[b]), nn.ReLU(True))
            for a, b in ((0, 4), (4, 6), (6, 8))])

        self.weight_initialization()
        self.dropout = nn.Dropout(0.3, True)

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, in_mod):
        x = in_mod['lidar']
        # points -> stride-1 voxels and all kernel maps up front (stem k3 at stride 1, then (k2 s2 down, k3) per encoder
        # stage): the host synchronisations of the forward; a trainer may hand them over prepared (in_mod['_geometry'])
        z, x0 = in_mod.get('_geometry') or prepare_geometry(x, self.pres, self.vres)
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)

        feats = [point_to_voxel(x0, z0)]
        for down in self.vox_downs:
            feats.append(down(feats[-1]))
        _, x1, x2, x3, x4 = feats

        z1 = voxel_to_point(x4, z0)
        z1.F = z1.F + self.point_transforms[0](z0.F)

        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self.vox_ups[0][0](y1)
        y1 = self.vox_ups[0][1](torchsparse.cat([y1, x3]))
        y2 = self.vox_ups[1][0](y1)
        y2 = self.vox_ups[1][1](torchsparse.cat([y2, x2]))
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + self.point_transforms[1](z1.F)

        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self.vox_ups[2][0](y3)
        y3 = self.vox_ups[2][1](torchsparse.cat([y3, x1]))
        y4 = self.vox_ups[3][0](y3)
        y4 = self.vox_ups[3][1](torchsparse.cat([y4, x0]))
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + self.point_transforms[2](z2.F)

        return {'x_vox': self.classifier_vox(z3.F)}
