"""CPU restatement of the SphereFormer block and the SPVCNN_SPFORMER network
(TEST INFRASTRUCTURE; rows a10, a12 of SURVEY.md §8a).

Follows core/models/sphereformer/spherical_transformer.py:12-28 (Mlp), :68-283
(SparseMultiheadSASphereConcat), :286-348 (SphereFormer) and
core/models/nuscenes/spvcnn_spformer.py:15-189 / spvcnn_swiftnet18_spformer_tsd_full.py:18-194
(teacher copy returning pts_feats) over oracle.torchsparse_cpu + oracle.sptr_ref.
The constructor reproduces the ``quant_size_sphere`` aliasing of SURVEY Appendix C-1:
every block's spherical quant size at forward time is the value left after the LAST
in-place scaling, while window sizes and cubic sizes are per block.
Pinned on golden vectors from the reference's own modules (tests/golden/make_golden.py).
"""
from functools import partial

import numpy as np
import torch
from torch import nn

from . import sptr_cpu as sptr
from . import torchsparse_cpu as ts
from .spvcnn_ref import (BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock, initial_voxelize,
                         point_to_voxel, voxel_to_point)
from .sptr_ref import cart2sphere, exponential_split
from .torchsparse_cpu import PointTensor
from .torchsparse_cpu import nn as spnn


class DropPath(nn.Module):
    """timm DropPath (stochastic depth over dim 0)."""

    def __init__(self, p=0.):
        super().__init__()
        self.drop_prob = p

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class SparseMultiheadSASphereConcat(nn.Module):
    def __init__(self, embed_dim, num_heads, window_size, window_size_sphere, quant_size, quant_size_sphere, a):
        super().__init__()
        self.num_heads = num_heads
        head_dim = embed_dim // num_heads
        self.scale = head_dim ** -0.5
        self.window_size = sptr.to_3d_numpy(window_size)
        self.window_size_sphere = sptr.to_3d_numpy(window_size_sphere)
        self.quant_size = sptr.to_3d_numpy(quant_size)
        self.quant_size_sphere = sptr.to_3d_numpy(quant_size_sphere)        # same object if ndarray (aliasing)
        self.a = a
        qgl = int((window_size[0] + 1e-4) / quant_size[0])
        h1 = num_heads // 2
        self.num_heads_brc1 = h1
        tn = nn.init.trunc_normal_
        self.relative_pos_query_table = nn.Parameter(tn(torch.zeros(2 * qgl - 1, 3, h1, head_dim), std=.02))
        self.relative_pos_key_table = nn.Parameter(tn(torch.zeros(2 * qgl - 1, 3, h1, head_dim), std=.02))
        self.relative_pos_value_table = nn.Parameter(tn(torch.zeros(2 * qgl - 1, 3, h1, head_dim), std=.02))
        self.quant_grid_length = qgl
        qgs = int((window_size_sphere[0] + 1e-4) / quant_size_sphere[0])
        h2 = num_heads - h1
        self.relative_pos_query_table_sphere = nn.Parameter(tn(torch.zeros(2 * qgs, 3, h2, head_dim), std=.02))
        self.relative_pos_key_table_sphere = nn.Parameter(tn(torch.zeros(2 * qgs, 3, h2, head_dim), std=.02))
        self.relative_pos_value_table_sphere = nn.Parameter(tn(torch.zeros(2 * qgs, 3, h2, head_dim), std=.02))
        self.quant_grid_length_sphere = qgs
        self.qkv = nn.Linear(embed_dim, embed_dim * 3, bias=True)
        self.proj = nn.Linear(embed_dim, embed_dim)

    def forward(self, feats, xyz, batch):
        N, C = feats.shape
        qkv = self.qkv(feats).reshape(N, 3, self.num_heads, C // self.num_heads).permute(1, 0, 2, 3).contiguous()
        query, key, value = qkv[0] * self.scale, qkv[1], qkv[2]
        xyz_sphere = cart2sphere(xyz)
        h1 = self.num_heads_brc1
        p1 = sptr.get_indices_params(xyz, batch, self.window_size, False)
        p2 = sptr.get_indices_params(xyz_sphere, batch, self.window_size_sphere, False)

        def run(sl, coords, p, window, quant, qgl, tabs, split):
            i0, i0o, n_max, i1, i1o, sort_idx = p
            return sptr.sparse_self_attention(
                query[:, sl].contiguous().float(), key[:, sl].contiguous().float(), value[:, sl].contiguous().float(),
                coords.float(), i0.int(), i0o.int(), n_max, i1.int(), i1o.int(), sort_idx, window, False,
                pe_type='contextual', rel_query=True, rel_key=True, rel_value=True, quant_size=quant,
                quant_grid_length=qgl, relative_pos_query_table=tabs[0].float(),
                relative_pos_key_table=tabs[1].float(), relative_pos_value_table=tabs[2].float(), split_func=split)

        out1 = run(slice(0, h1), xyz, p1, self.window_size, self.quant_size, self.quant_grid_length,
                   (self.relative_pos_query_table, self.relative_pos_key_table, self.relative_pos_value_table), None)
        out2 = run(slice(h1, None), xyz_sphere, p2, self.window_size_sphere, self.quant_size_sphere,
                   self.quant_grid_length_sphere,
                   (self.relative_pos_query_table_sphere, self.relative_pos_key_table_sphere,
                    self.relative_pos_value_table_sphere), partial(exponential_split, a=self.a))
        x = torch.cat([out1, out2], 1).view(N, C)
        return self.proj(x)


class SphereFormer(nn.Module):
    def __init__(self, dim, num_heads, window_size, window_size_sphere, quant_size, quant_size_sphere, drop_path=0.0,
                 a=0.0125):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = SparseMultiheadSASphereConcat(dim, num_heads, window_size, window_size_sphere, quant_size,
                                                  quant_size_sphere, a)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * 4.0))

    def forward(self, feats, xyz, batch):
        short_cut = feats
        feats = self.attn(self.norm1(feats), xyz, batch)
        feats = short_cut + self.drop_path(feats)
        return feats + self.drop_path(self.mlp(self.norm2(feats)))


class SPVCNN_SPFORMER(nn.Module):
    def __init__(self, cr, in_channel, num_classes, window_size, window_size_sphere, quant_size, quant_size_sphere,
                 window_size_scale, drop_path_rate, a, pres, vres, return_pts_feats=False):
        super().__init__()
        cs = [int(cr * x) for x in [32, 32, 64, 128, 256, 256, 128, 96, 96]]
        self.pres, self.vres = pres, vres
        self.return_pts_feats = return_pts_feats
        self.stem = nn.Sequential(
            spnn.Conv3d(in_channel, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))
        self.vox_downs = nn.ModuleList()
        for idx in range(4):
            self.vox_downs.append(nn.Sequential(
                BasicConvolutionBlock(cs[idx], cs[idx], ks=2, stride=2, dilation=1),
                ResidualBlock(cs[idx], cs[idx + 1], ks=3, stride=1, dilation=1),
                ResidualBlock(cs[idx + 1], cs[idx + 1], ks=3, stride=1, dilation=1)))
        self.window_size = window_size
        self.window_size_sphere = window_size_sphere
        self.quant_size = quant_size
        self.quant_size_sphere = quant_size_sphere
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 7)]
        self.transformer_blocks = nn.ModuleList()
        for idx in range(1, 5):
            self.transformer_blocks.append(SphereFormer(
                cs[idx], cs[idx] // 16, self.window_size, self.window_size_sphere, self.quant_size,
                self.quant_size_sphere, drop_path=dpr[idx], a=a))
            sc, ss = window_size_scale
            self.window_size = self.window_size * sc
            self.quant_size = self.quant_size * sc
            self.window_size_sphere[0] = self.window_size_sphere[0] * ss
            self.window_size_sphere[1] = self.window_size_sphere[1] * ss
            self.quant_size_sphere[0] = self.quant_size_sphere[0] * ss       # in place: aliased by every block
            self.quant_size_sphere[1] = self.quant_size_sphere[1] * ss
        self.vox_ups = nn.ModuleList()
        for idx in range(4, len(cs) - 1):
            self.vox_ups.append(nn.ModuleList([
                BasicDeconvolutionBlock(cs[idx], cs[idx + 1], ks=2, stride=2),
                nn.Sequential(
                    ResidualBlock(cs[idx + 1] + cs[len(cs) - 1 - (1 + idx)], cs[idx + 1], ks=3, stride=1, dilation=1),
                    ResidualBlock(cs[idx + 1], cs[idx + 1], ks=3, stride=1, dilation=1))]))
        self.classifier_vox = nn.Sequential(nn.Linear(cs[8], num_classes))
        self.point_transforms = nn.ModuleList([
            nn.Sequential(nn.Linear(cs[0], cs[4]), nn.BatchNorm1d(cs[4]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[4], cs[6]), nn.BatchNorm1d(cs[6]), nn.ReLU(True)),
            nn.Sequential(nn.Linear(cs[6], cs[8]), nn.BatchNorm1d(cs[8]), nn.ReLU(True))])
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.dropout = nn.Dropout(0.3, True)

    def forward(self, in_mod):
        x = in_mod['lidar']
        z = PointTensor(x.F, x.C.float())
        x0 = initial_voxelize(z, self.pres, self.vres)
        zz = PointTensor(x0.F, x0.C.float())
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)
        vox_feats = [point_to_voxel(x0, z0)]
        pts_feats = []
        for idx, vox_block in enumerate(self.vox_downs):
            vox_out = vox_block(vox_feats[idx])
            tmp_p = point_to_voxel(vox_out, zz)
            coord_xyz, batch = tmp_p.F[:, :3], tmp_p.C[:, 3]
            vox_out.F = self.transformer_blocks[idx](vox_out.F, coord_xyz, batch)
            vox_feats.append(vox_out)
            if idx == 3 and self.return_pts_feats:
                pts_feats.append(voxel_to_point(vox_out, z0).F)
        x1, x2, x3, x4 = vox_feats[1:5]
        z1 = voxel_to_point(x4, z0)
        z1.F = z1.F + self.point_transforms[0](z0.F)
        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self.vox_ups[0][0](y1)
        y1 = self.vox_ups[0][1](ts.cat([y1, x3]))
        y2 = self.vox_ups[1][0](y1)
        y2 = self.vox_ups[1][1](ts.cat([y2, x2]))
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + self.point_transforms[1](z1.F)
        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self.vox_ups[2][0](y3)
        y3 = self.vox_ups[2][1](ts.cat([y3, x1]))
        y4 = self.vox_ups[3][0](y3)
        y4 = self.vox_ups[3][1](ts.cat([y4, x0]))
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + self.point_transforms[2](z2.F)
        out = {'x_vox': self.classifier_vox(z3.F)}
        if self.return_pts_feats:
            out['pts_feats'] = pts_feats
        return out


def default_spformer_kwargs(voxel_size=0.05, cr=1.0, in_channel=4, num_classes=17, drop_path_rate=0.3):
    """What core/builder.py:533-554 passes for configs/nuscenes/train/spformer.yaml
    (patch_size 1, window_size 6, quant_size_scale 24, window_size_sphere [2,2,120],
    window_size_scale [2,2], a 0.0125)."""
    patch = np.array([voxel_size * 1] * 3).astype(np.float32)
    window = patch * 6
    wss = [2, 2, 120]
    return dict(cr=cr, in_channel=in_channel, num_classes=num_classes, window_size=window, window_size_sphere=wss,
                quant_size=window / 24, quant_size_sphere=np.array(wss) / 24, window_size_scale=[2.0, 2.0],
                drop_path_rate=drop_path_rate, a=0.0125, pres=voxel_size, vres=voxel_size)


def over(package, sptr_package):
    """TESTS ONLY (see spvcnn_ref.over): a second copy of this module over another torchsparse-shaped operator package and
    another sptr-shaped attention package (``to_3d_numpy, get_indices_params, sparse_self_attention`` with the signatures of
    third_party/SparseTransformer/sptr/{utils,modules}.py) -- the reference's SPVCNN_SPFORMER call sequence
    (spvcnn_spformer.py:125-189, spherical_transformer.py:165-348) over the product's operators on the GPU."""
    import importlib
    import importlib.util
    import sys
    from . import spvcnn_ref
    name = __name__ + '__over__' + package.__name__.replace('.', '_')
    if name in sys.modules:
        return sys.modules[name]
    base = spvcnn_ref.over(package)
    spec = importlib.util.spec_from_file_location(name, __file__)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = __package__
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.ts, mod.sptr = package, sptr_package
    mod.PointTensor = package.PointTensor
    mod.spnn = importlib.import_module(package.__name__ + '.nn')
    for n in ('BasicConvolutionBlock', 'BasicDeconvolutionBlock', 'ResidualBlock', 'initial_voxelize', 'point_to_voxel',
              'voxel_to_point'):
        setattr(mod, n, getattr(base, n))
    return mod
