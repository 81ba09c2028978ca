"""SphereFormer block (row a10; core/models/sphereformer/spherical_transformer.py:12-348)
on the fused HIP window attention of ``u2mkd_amd.sptr``.

Half of the heads attend inside cubic xyz windows, the other half inside spherical
(theta, beta, r) windows with an exponentially split radial index; both use contextual
relative-position tables for query, key and value.  Parameter names match the reference
(``norm1``, ``attn.qkv``, ``attn.relative_pos_*_table[_sphere]``, ``attn.proj``, ``norm2``,
``mlp.fc1/fc2``) so its checkpoints load unchanged.
"""
import os

import numpy as np
import torch
from torch import nn

from .. import sptr
from .blocks import PointLinear

_PACKED = os.environ.get('U2MKD_SPTR_PACKED', '1') != '0'     # 0: slice / scale / concatenate around the contiguous kernels

__all__ = ['SphereFormer', 'SparseMultiheadSASphereConcat', 'DropPath', 'cart2sphere']


def cart2sphere(xyz):
    """(theta [0,360) deg, beta [0,180] deg, r) -- spherical_transformer.py:31-36."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    theta = (torch.atan2(y, x) + np.pi) * 180 / np.pi
    beta = torch.atan2(torch.sqrt(x ** 2 + y ** 2), z) * 180 / np.pi
    r = torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    return torch.stack([theta, beta, r], -1)


_cart2sphere_torch = cart2sphere       # (a caller that substitutes `cart2sphere` -- the CPU-libm fixture of tests/test_kd_path.py --
                                       # gets its function: the fused plan preparation stands in for THIS formulation only)


class DropPath(nn.Module):
    """Stochastic depth over dim 0 (timm.models.layers.DropPath, as the reference uses it)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * (mask / keep)              # (one pass over x, forward and backward; timm: x.div(keep) * mask)

    def add_to(self, shortcut, x):
        """``shortcut + self(x)`` -- the residual form every use in SphereFormer has -- with the add inside the masked
        product's pass (one launch fewer per use; the same values: shortcut + x * (mask / keep))."""
        if self.drop_prob == 0. or not self.training:
            return shortcut + x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return torch.addcmul(shortcut, x, mask / keep)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        # nn.Linear of the reference (same parameters and state-dict keys) on this package's MFMA pipeline: forward /
        # input gradient on the pair kernel's dense mode, the weight gradient (a [out, N] x [N, in] product whose
        # reduction runs over all N tokens) on the pair-list kernel with its fixed-order slab sum -- rocBLAS answers
        # that shape with an atomic split-K kernel (not reproducible from run to run) or, atomics off, ONE serial tile
        # (14 ms per product at N = 80 000, measured)
        self.fc1 = PointLinear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = PointLinear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class SparseMultiheadSASphereConcat(nn.Module):
    def __init__(self, embed_dim, num_heads, window_size, window_size_sphere, quant_size, quant_size_sphere, a):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        head_dim = embed_dim // num_heads
        assert head_dim == 16, 'the sptr kernels are built for head_dim 16 (sptr/functional.py:355)'
        self.scale = head_dim ** -0.5
        # to_3d_numpy keeps ndarray identity: quant_size_sphere is shared (and later mutated) by the
        # caller exactly as in the reference
        self.window_size = sptr.to_3d_numpy(window_size)
        self.window_size_sphere = sptr.to_3d_numpy(window_size_sphere)
        self.quant_size = sptr.to_3d_numpy(quant_size)
        self.quant_size_sphere = sptr.to_3d_numpy(quant_size_sphere)
        self.a = a
        qgl = int((window_size[0] + 1e-4) / quant_size[0])
        assert qgl == int((window_size[1] + 1e-4) / quant_size[1])
        h1 = num_heads // 2
        h2 = num_heads - h1
        self.num_heads_brc1 = h1
        self.quant_grid_length = qgl

        def table(rows, heads):
            return nn.Parameter(nn.init.trunc_normal_(torch.zeros(rows, 3, heads, head_dim), std=.02))

        self.relative_pos_query_table = table(2 * qgl - 1, h1)
        self.relative_pos_key_table = table(2 * qgl - 1, h1)
        self.relative_pos_value_table = table(2 * qgl - 1, h1)
        qgs = int((window_size_sphere[0] + 1e-4) / quant_size_sphere[0])
        assert qgs == int((window_size_sphere[1] + 1e-4) / quant_size_sphere[1])
        self.quant_grid_length_sphere = qgs
        self.relative_pos_query_table_sphere = table(2 * qgs, h2)
        self.relative_pos_key_table_sphere = table(2 * qgs, h2)
        self.relative_pos_value_table_sphere = table(2 * qgs, h2)
        self.qkv = PointLinear(embed_dim, embed_dim * 3, bias=True)
        self.proj = PointLinear(embed_dim, embed_dim)

    def plans(self, xyz, batch, quantised=False):
        """(cubic window plan, spherical window plan, spherical coordinates) of the tokens ``xyz`` [N, 3] float / ``batch`` [N]:
        everything of the attention that depends on the token POSITIONS only (spherical_transformer.py:31-36, 206-213).  On
        the device the triple is cached on ``xyz`` (spf._plan), so a caller that knows the positions ahead of the features -- a
        trainer preparing the next batch, kd.py ``prefetch_plans`` -- builds it there (``quantised``: and the quantised
        in-window coordinates both branches look their tables up with) and the forward finds it."""
        if xyz.is_cuda and cart2sphere is _cart2sphere_torch:
            from ..torchsparse.nn import functional as spf
            key = 'sptr_pair_%r_%r' % (tuple(float(v) for v in self.window_size), tuple(float(v) for v in self.window_size_sphere))
            plan, plan_s, xyz_sphere = spf._plan(
                xyz, key, lambda: sptr.WindowPlan.pair(xyz, batch, self.window_size, self.window_size_sphere), batch)
        else:
            xyz_sphere = cart2sphere(xyz)
            plan = sptr.WindowPlan(xyz, batch, self.window_size)
            plan_s = sptr.WindowPlan(xyz_sphere, batch, self.window_size_sphere)
        if quantised:
            plan.quant_coords(xyz, self.quant_size, False)
            plan_s.quant_coords(xyz_sphere, self.quant_size_sphere, True)
        return plan, plan_s, xyz_sphere

    def forward(self, feats, xyz, batch):
        N, C = feats.shape
        qkv = self.qkv(feats).reshape(N, 3, self.num_heads, C // self.num_heads)
        h1 = self.num_heads_brc1
        xyz = xyz.float()
        plan, plan_s, xyz_sphere = self.plans(xyz, batch)
        cubic = (0, h1, xyz, plan, self.quant_size, self.quant_grid_length,
                 (self.relative_pos_query_table, self.relative_pos_key_table, self.relative_pos_value_table), None)
        sphere = (h1, self.num_heads - h1, xyz_sphere, plan_s, self.quant_size_sphere, self.quant_grid_length_sphere,
                  (self.relative_pos_query_table_sphere, self.relative_pos_key_table_sphere,
                   self.relative_pos_value_table_sphere), self.a)
        if qkv.is_cuda and C // self.num_heads == 16 and _PACKED:
            # q = qkv[:, 0] * scale, the per-branch head slices and torch.cat([out1, out2], 1) (spherical_transformer.py
            # :192-228) all inside the two kernels, through row strides
            x = sptr.packed_window_attention(qkv, self.scale, [cubic, sphere])
        else:
            query = qkv[:, 0] * self.scale
            key, value = qkv[:, 1], qkv[:, 2]
            outs = [sptr.window_attention(query[:, h0:h0 + h], key[:, h0:h0 + h], value[:, h0:h0 + h], pts, pl, qs, qgl, *tabs, a)
                    for h0, h, pts, pl, qs, qgl, tabs, a in (cubic, sphere)]
            x = torch.cat(outs, 1).view(N, C)
        return self.proj(x)


class SphereFormer(nn.Module):
    def __init__(self, dim, num_heads, window_size, window_size_sphere, quant_size, quant_size_sphere,
                 indice_key=None, pe_type='contextual', rel_query=True, rel_key=True, rel_value=True, drop_path=0.0,
                 mlp_ratio=4.0, a=0.05 * 0.25):
        super().__init__()
        assert pe_type == 'contextual' and rel_query and rel_key and rel_value
        self.window_size = window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = SparseMultiheadSASphereConcat(dim, num_heads, window_size, window_size_sphere, quant_size,
                                                  quant_size_sphere, a)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, feats, xyz, batch):
        short_cut = feats
        feats = self.attn(self.norm1(feats), xyz, batch)
        if isinstance(self.drop_path, DropPath):
            feats = self.drop_path.add_to(short_cut, feats)
            return self.drop_path.add_to(feats, self.mlp(self.norm2(feats)))
        feats = short_cut + self.drop_path(feats)
        return feats + self.drop_path(self.mlp(self.norm2(feats)))
