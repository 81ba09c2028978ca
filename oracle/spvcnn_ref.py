"""CPU restatement of the reference SPVCNN and its point/voxel helpers
(TEST INFRASTRUCTURE; rows a1-a3, a9 of SURVEY.md §8a).

Follows core/models/utils.py:15-118, core/models/build_blocks.py:21-83 and
core/models/semantickitti/spvcnn.py:10-142 line by line over
``oracle.torchsparse_cpu``.  Pinned in the build container against the
reference's own SPVCNN class imported over the same CPU operators
(tests/golden/make_golden.py -> tests/golden/spvcnn_cr05_*.npz): both must give
bit-identical logits for the same state dict.
"""
import torch
from torch import nn

from . import torchsparse_cpu as ts
from .torchsparse_cpu import PointTensor, SparseTensor
from .torchsparse_cpu import nn as spnn
from .torchsparse_cpu.nn import functional as spf
from .torchsparse_cpu.nn.utils import get_kernel_offsets


def initial_voxelize(z, init_res, after_res):  # utils.py:15-35
    new_float_coord = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
    pc_hash = spf.sphash(torch.floor(new_float_coord).int())
    sparse_hash = torch.unique(pc_hash)
    idx_query = spf.sphashquery(pc_hash, sparse_hash)
    counts = spf.spcount(idx_query.int(), len(sparse_hash))
    inserted_coords = spf.spvoxelize(torch.floor(new_float_coord), idx_query, counts)
    inserted_coords = torch.round(inserted_coords).int()
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.cmaps.setdefault(new_tensor.stride, new_tensor.coords)
    z.additional_features['idx_query'][1] = idx_query
    z.additional_features['counts'][1] = counts
    z.C = new_float_coord
    return new_tensor


def point_to_voxel(x, z):  # utils.py:40-65
    if z.additional_features is None or z.additional_features.get('idx_query') is None \
            or z.additional_features['idx_query'].get(x.s) is None:
        pc_hash = spf.sphash(torch.cat([torch.floor(z.C[:, :3] / x.s[0]).int() * x.s[0],
                                        z.C[:, -1].int().view(-1, 1)], 1))
        sparse_hash = spf.sphash(x.C)
        idx_query = spf.sphashquery(pc_hash, sparse_hash)
        counts = spf.spcount(idx_query.int(), x.C.shape[0])
        z.additional_features['idx_query'][x.s] = idx_query
        z.additional_features['counts'][x.s] = counts
    else:
        idx_query = z.additional_features['idx_query'][x.s]
        counts = z.additional_features['counts'][x.s]
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(inserted_feat, x.C, x.s)
    new_tensor.cmaps = x.cmaps
    new_tensor.kmaps = x.kmaps
    return new_tensor


def voxel_to_point(x, z, nearest=False):  # utils.py:70-118
    if z.idx_query is None or z.weights is None or z.idx_query.get(x.s) is None \
            or z.weights.get(x.s) is None:
        off = get_kernel_offsets(2, x.s, 1, device=z.F.device)
        old_hash = spf.sphash(torch.cat([torch.floor(z.C[:, :3] / x.s[0]).int() * x.s[0],
                                         z.C[:, -1].int().view(-1, 1)], 1), off)
        pc_hash = spf.sphash(x.C.to(z.F.device))
        idx_query = spf.sphashquery(old_hash, pc_hash)
        weights = spf.calc_ti_weights(z.C, idx_query, scale=x.s[0]).transpose(0, 1).contiguous()
        idx_query = idx_query.transpose(0, 1).contiguous()
        if nearest:
            weights[:, 1:] = 0.
            idx_query[:, 1:] = -1
        new_feat = spf.spdevoxelize(x.F, idx_query, weights)
        new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
        new_tensor.additional_features = z.additional_features
        new_tensor.idx_query[x.s] = idx_query
        new_tensor.weights[x.s] = weights
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
    else:
        new_feat = spf.spdevoxelize(x.F, z.idx_query.get(x.s), z.weights.get(x.s))
        new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
        new_tensor.additional_features = z.additional_features
    return new_tensor


class BasicConvolutionBlock(nn.Module):  # build_blocks.py:21-36
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
                                 spnn.BatchNorm(outc), spnn.ReLU(True))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):  # build_blocks.py:39-52
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, transposed=True),
                                 spnn.BatchNorm(outc), spnn.ReLU(True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):  # build_blocks.py:55-83
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
            spnn.BatchNorm(outc), spnn.ReLU(True),
            spnn.Conv3d(outc, outc, kernel_size=ks, dilation=dilation, stride=1),
            spnn.BatchNorm(outc))
        self.downsample = nn.Sequential() if (inc == outc and stride == 1) else nn.Sequential(
            spnn.Conv3d(inc, outc, kernel_size=1, dilation=1, stride=stride), spnn.BatchNorm(outc))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        return self.relu(self.net(x) + self.downsample(x))


class SPVCNN(nn.Module):  # semantickitti/spvcnn.py:10-142
    def __init__(self, **kwargs):
        super().__init__()
        cr = kwargs.get('cr')
        cs = [int(cr * x) for x in [32, 32, 64, 128, 256, 256, 128, 96, 96]]
        self.in_channel = kwargs.get('in_channel', 4)
        self.num_classes = kwargs.get('num_classes', 17)
        self.pres = kwargs.get('pres')
        self.vres = kwargs.get('vres')
        self.stem = nn.Sequential(
            spnn.Conv3d(self.in_channel, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))
        self.vox_downs = nn.ModuleList()
        for idx in range(4):
            self.vox_downs.append(nn.Sequential(
                BasicConvolutionBlock(cs[idx], cs[idx], ks=2, stride=2, dilation=1),
                ResidualBlock(cs[idx], cs[idx + 1], ks=3, stride=1, dilation=1),
                ResidualBlock(cs[idx + 1], cs[idx + 1], ks=3, stride=1, dilation=1)))
        self.vox_ups = nn.ModuleList()
        for idx in range(4, len(cs) - 1):
            self.vox_ups.append(nn.ModuleList([
                BasicDeconvolutionBlock(cs[idx], cs[idx + 1], ks=2, stride=2),
                nn.Sequential(
                    ResidualBlock(cs[idx + 1] + cs[len(cs) - 1 - (1 + idx)], cs[idx + 1], ks=3, stride=1, dilation=1),
                    ResidualBlock(cs[idx + 1], cs[idx + 1], ks=3, stride=1, dilation=1))]))
        self.classifier_vox = nn.Sequential(nn.Linear(cs[8], self.num_classes))
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
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)
        vox_feats = [point_to_voxel(x0, z0)]
        for idx, vox_block in enumerate(self.vox_downs):
            vox_feats.append(vox_block(vox_feats[idx]))
        x1, x2, x3, x4 = vox_feats[1:5]
        z1 = voxel_to_point(x4, z0)
        z1.F = z1.F + self.point_transforms[0](z0.F)
        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self.vox_ups[0][0](y1)
        y1 = ts.cat([y1, x3])
        y1 = self.vox_ups[0][1](y1)
        y2 = self.vox_ups[1][0](y1)
        y2 = ts.cat([y2, x2])
        y2 = self.vox_ups[1][1](y2)
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + self.point_transforms[1](z1.F)
        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self.vox_ups[2][0](y3)
        y3 = ts.cat([y3, x1])
        y3 = self.vox_ups[2][1](y3)
        y4 = self.vox_ups[3][0](y3)
        y4 = ts.cat([y4, x0])
        y4 = self.vox_ups[3][1](y4)
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + self.point_transforms[2](z2.F)
        return {'x_vox': self.classifier_vox(z3.F)}


def lovasz_grad(gt_sorted):  # criterions.py:40-52
    p = len(gt_sorted)
    gts = gt_sorted.sum()
    intersection = gts - gt_sorted.float().cumsum(0)
    union = gts + (1 - gt_sorted).float().cumsum(0)
    jaccard = 1. - intersection / union
    if p > 1:
        jaccard[1:p] = jaccard[1:p] - jaccard[0:-1]
    return jaccard


def lovasz_softmax_flat(probas, labels):  # criterions.py:73-101, classes='present'
    if probas.numel() == 0:
        return probas * 0.
    C = probas.size(1)
    losses = []
    for c in range(C):
        fg = (labels == c).float()
        if fg.sum() == 0:
            continue
        errors = (fg - probas[:, c]).abs()
        errors_sorted, perm = torch.sort(errors, 0, descending=True)
        losses.append(torch.dot(errors_sorted, lovasz_grad(fg[perm.data])))
    if not losses:
        return 0
    return sum(losses) / len(losses)


def mix_lovasz_cross_entropy(x, y, ignore_index=0):  # criterions.py:159-174 + flatten_probas :129-146
    probas = torch.softmax(x, 1)
    valid = y != ignore_index
    vprobas = probas[valid.nonzero().squeeze()]
    vlabels = y[valid]
    lov = lovasz_softmax_flat(vprobas, vlabels)
    ce = nn.functional.cross_entropy(x, y, ignore_index=ignore_index)
    return lov + ce


def fill_state_by_name(model, seed=0, conv2d_he=False):
    """Deterministic, construction-order-independent parameter fill: every
    tensor of the state dict is drawn from a generator seeded by its KEY, so
    the reference class, this restatement and the HIP model get identical
    weights without shipping a checkpoint.  ``conv2d_he``: the camera branch's Conv2d (4-D) and the fusion blocks' Conv1d (3-D) weights get a zero-mean
    He-scaled fill instead of the positive BatchNorm-gamma fill they share by default -- needed when BatchNorm runs on
    its running statistics (eval mode): an all-positive 3x3xC filter multiplies the scale by ~9C per layer, which
    batch statistics undo and running statistics do not."""
    import zlib
    sd = model.state_dict()
    out = {}
    for key, t in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) % (2 ** 31))
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros_like(t)
        elif key.endswith('running_var'):
            out[key] = 0.5 + torch.rand(t.shape, generator=g)
        elif key.endswith('running_mean'):
            out[key] = 0.1 * torch.randn(t.shape, generator=g)
        elif key.endswith('.kernel'):
            fan = t.shape[-2] * (t.shape[0] if t.dim() == 3 else 1)
            out[key] = torch.randn(t.shape, generator=g) * (2.0 / fan) ** 0.5
        elif t.dim() in (3, 4) and conv2d_he:  # Conv2d weight [out, in, kh, kw] / Conv1d weight [out, in, k]
            out[key] = torch.randn(t.shape, generator=g) * (2.0 / t[0].numel()) ** 0.5
        elif t.dim() == 2:  # nn.Linear weight [out, in]
            out[key] = torch.randn(t.shape, generator=g) * (1.0 / t.shape[1]) ** 0.5
        elif key.endswith('weight'):  # BN gamma
            out[key] = 0.8 + 0.4 * torch.rand(t.shape, generator=g)
        else:  # biases
            out[key] = 0.05 * torch.randn(t.shape, generator=g)
    model.load_state_dict(out)
    return model


def over(package):
    """TESTS ONLY: a second copy of this module whose operator package is ``package`` -- anything with the torchsparse
    v1.4.0 surface this file uses (``SparseTensor / PointTensor / cat``, ``nn.{Conv3d, BatchNorm, ReLU}``,
    ``nn.functional.{sphash, sphashquery, spcount, spvoxelize, spdevoxelize, calc_ti_weights}``,
    ``nn.utils.get_kernel_offsets``).  With ``u2mkd_amd.torchsparse`` on the GPU box this is the REFERENCE'S OWN CALL
    SEQUENCE (core/models/utils.py:15-118, build_blocks.py:21-83, spvcnn.py:85-142: un-fused Conv3d -> BatchNorm -> ReLU
    modules, lazily built kernel maps, torch.unique) running over the product's operators, where /root/reference does
    not exist (tests/test_gpu_reference_sequence.py).  The module-level operator names are looked up at call time, so
    re-binding them in the copy is enough."""
    import importlib
    import importlib.util
    import sys
    name = __name__ + '__over__' + package.__name__.replace('.', '_')
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, __file__)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = __package__
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.ts = package
    mod.PointTensor, mod.SparseTensor = package.PointTensor, package.SparseTensor
    mod.spnn = importlib.import_module(package.__name__ + '.nn')
    mod.spf = importlib.import_module(package.__name__ + '.nn.functional')
    mod.get_kernel_offsets = importlib.import_module(package.__name__ + '.nn.utils').get_kernel_offsets
    return mod
