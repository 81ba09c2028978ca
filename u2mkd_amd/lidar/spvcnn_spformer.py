"""SPVCNN + SphereFormer LiDAR network (rows a10, a12): the uni-modal teacher of the KD
pipeline (core/models/nuscenes/spvcnn_spformer.py:15-189; its copy in
spvcnn_swiftnet18_spformer_tsd_full.py:18-194 additionally returns ``pts_feats``).

The reference reads its hyper-parameters from the global torchpack ``configs``; here they are
explicit keyword arguments (see :func:`spformer_kwargs` for what core/builder.py:533-554
passes).  The constructor keeps the reference's parameter flow, including the in-place
scaling of the shared ``quant_size_sphere`` array."""
import numpy as np
import torch
from torch import nn

from .. import torchsparse
from ..torchsparse import PointTensor
from ..torchsparse import nn as spnn
from ..torchsparse.nn import functional as spf
from .blocks import (BasicConvolutionBlock, BasicDeconvolutionBlock, FusedSequential, PointBatchNorm1d, PointLinear,
                     ResidualBlock)
from .point_voxel import initial_voxelize, point_to_voxel, prepare_geometry, voxel_to_point
from .sphereformer import SphereFormer

__all__ = ['SPVCNN_SPFORMER', 'spformer_kwargs']

_BASE_CHANNELS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


def spformer_kwargs(voxel_size=0.05, cr=1.0, in_channel=4, num_classes=17, drop_path_rate=0.3, patch_size=1,
                    window_size=6, quant_size_scale=24, window_size_sphere=(2, 2, 120), window_size_scale=(2.0, 2.0),
                    a=0.0125):
    """Arguments core/builder.py:533-554 builds from configs/nuscenes/train/spformer.yaml."""
    patch = np.array([voxel_size * patch_size] * 3).astype(np.float32)
    window = patch * window_size
    wss = list(window_size_sphere)
    return dict(cr=cr, in_channel=in_channel, num_classes=num_classes, window_size=window, window_size_sphere=wss,
                quant_size=window / quant_size_scale, quant_size_sphere=np.array(wss) / quant_size_scale,
                window_size_scale=list(window_size_scale), drop_path_rate=drop_path_rate, a=a, pres=voxel_size,
                vres=voxel_size)


class SPVCNN_SPFORMER(nn.Module):
    def __init__(self, cr, in_channel, num_classes, window_size, window_size_sphere, quant_size, quant_size_sphere,
                 window_size_scale, drop_path_rate, a, pres, vres, return_pts_feats=False):
        super().__init__()
        cs = [int(cr * c) for c in _BASE_CHANNELS]
        self.cs = cs
        self.in_channel, self.num_classes, self.out_channel = in_channel, num_classes, cs[-1]
        self.pres, self.vres = pres, vres
        self.return_pts_feats = return_pts_feats

        self.stem = FusedSequential(
            spnn.Conv3d(in_channel, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))
        self.vox_downs = nn.ModuleList([
            nn.Sequential(BasicConvolutionBlock(cs[i], cs[i], ks=2, stride=2, dilation=1),
                          ResidualBlock(cs[i], cs[i + 1], ks=3, stride=1, dilation=1),
                          ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))
            for i in range(4)])

        self.window_size = window_size
        self.window_size_sphere = window_size_sphere
        self.quant_size = quant_size
        self.quant_size_sphere = quant_size_sphere
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 7)]
        self.transformer_blocks = nn.ModuleList()
        for idx in range(1, 5):
            self.transformer_blocks.append(SphereFormer(
                cs[idx], cs[idx] // 16, self.window_size, self.window_size_sphere, self.quant_size,
                self.quant_size_sphere, indice_key='sphereformer{}'.format(idx + 1), drop_path=dpr[idx], a=a))
            scale_cubic, scale_sphere = window_size_scale
            self.window_size = self.window_size * scale_cubic          # re-bound: per-block values
            self.quant_size = self.quant_size * scale_cubic
            self.window_size_sphere[0] = self.window_size_sphere[0] * scale_sphere
            self.window_size_sphere[1] = self.window_size_sphere[1] * scale_sphere
            self.quant_size_sphere[0] = self.quant_size_sphere[0] * scale_sphere   # in place: seen by every block
            self.quant_size_sphere[1] = self.quant_size_sphere[1] * scale_sphere

        self.vox_ups = nn.ModuleList([
            nn.ModuleList([
                BasicDeconvolutionBlock(cs[i], cs[i + 1], ks=2, stride=2),
                nn.Sequential(
                    ResidualBlock(cs[i + 1] + cs[len(cs) - 2 - i], cs[i + 1], ks=3, stride=1, dilation=1),
                    ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))])
            for i in range(4, len(cs) - 1)])
        self.classifier_vox = nn.Sequential(PointLinear(cs[8], num_classes))
        self.point_transforms = nn.ModuleList([
            FusedSequential(PointLinear(cs[a_], cs[b_]), PointBatchNorm1d(cs[b_]), nn.ReLU(True))
            for a_, b_ in ((0, 4), (4, 6), (6, 8))])
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.dropout = nn.Dropout(0.3, True)

    def forward(self, in_mod):
        x = in_mod['lidar']
        # (z, x0): points -> stride-1 voxels + every kernel map of the encoder; prepared ahead by the trainer
        # (in_mod['_geometry'], point_voxel.prepare_geometry) or here -- the only host synchronisations of the forward
        z, x0 = in_mod.get('_geometry') or prepare_geometry(x, self.pres, self.vres)
        zz = PointTensor(x0.F, x0.C.float())          # carries the metric xyz of every stride-1 voxel
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)

        feats = [point_to_voxel(x0, z0)]
        pts_feats = []
        for idx, down in enumerate(self.vox_downs):
            vox_out = down(feats[idx])
            tmp_p = point_to_voxel(vox_out, zz)       # mean metric xyz (+ intensity) per coarse voxel
            coord_xyz, batch = tmp_p.F[:, :3].contiguous(), tmp_p.C[:, 3]      # (one copy: the plan kernels need contiguous rows)
            vox_out.F = self.transformer_blocks[idx](vox_out.F, coord_xyz, batch)
            feats.append(vox_out)
            if idx == 3 and self.return_pts_feats:
                pts_feats.append(voxel_to_point(vox_out, z0).F)
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

        out = {'x_vox': self.classifier_vox(z3.F)}
        if self.return_pts_feats:
            out['pts_feats'] = pts_feats
        return out
