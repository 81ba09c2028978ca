"""Point <-> voxel transfers (rows a1-a3 of SURVEY.md §8a).

Same results and the same caching keys as the reference's
``core/models/utils.py`` (initial_voxelize :15-35, point_to_voxel :40-65,
voxel_to_point :70-118, fetch_idx :121-135, SparseSyncBatchNorm :138-220), but
each step is one HIP launch: the 8-corner hash + probe is fused with the
trilinear weights and the [8,N]->[N,8] transposes, and index tensors are kept
in the int32 form the kernels consume.
"""
import os

import torch
from torch import nn

from .. import torchsparse
from ..torchsparse import PointTensor, SparseTensor
from ..torchsparse import nn as spnn
from ..torchsparse.nn import functional as spf
from ..torchsparse.nn.utils import fapply, get_kernel_offsets

__all__ = ['initial_voxelize', 'point_to_voxel', 'voxel_to_point', 'fetch_idx', 'SparseSyncBatchNorm']


def _floor_coords(pc, stride):
    """int32 (floor(xyz / s) * s, b) of float point coords [N,4]."""
    if pc.is_cuda and pc.dtype == torch.float32 and pc.shape[1] == 4:
        from .. import _lib as L
        pc = pc.contiguous()
        out = torch.empty(pc.shape[0], 4, dtype=torch.int32, device=pc.device)
        L.call('u2mkd_floor_coords', L.ptr(pc), pc.shape[0], int(stride), L.ptr(out), L.stream())
        return out
    xyz = torch.floor(pc[:, :3] / stride).int() * stride
    return torch.cat([xyz, pc[:, -1].int().view(-1, 1)], 1)


def _voxelize_issue(z: PointTensor, init_res, after_res):
    """First half of initial_voxelize: everything up to the (deferred) unique of the point hashes."""
    new_float_coord = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
    floor_c = torch.floor(new_float_coord)
    pc_hash = spf.sphash(floor_c.int())
    buf, cnt = spf.unique_sorted_deferred(pc_hash)
    return dict(z=z, coord=new_float_coord, floor_c=floor_c, pc_hash=pc_hash, buf=buf, cnt=cnt)


def _voxelize_finish(st, n_vox) -> SparseTensor:
    z, floor_c = st['z'], st['floor_c']
    sparse_hash = st['buf'][:n_vox]
    idx_query = spf.sphashquery(st['pc_hash'], sparse_hash)
    counts = spf.spcount(idx_query.int(), n_vox)

    inserted_coords = spf.spvoxelize(floor_c, idx_query, counts)
    inserted_coords = torch.round(inserted_coords).int()
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts)

    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.cmaps.setdefault(new_tensor.stride, new_tensor.coords)
    z.additional_features['idx_query'][1] = idx_query
    z.additional_features['counts'][1] = counts
    z.C = st['coord']
    return new_tensor


def initial_voxelize(z: PointTensor, init_res, after_res) -> SparseTensor:
    """Points -> stride-1 voxels (mean of coords and feats per voxel); voxel
    order = ascending FNV hash (the sorted unique hashes), as core/models/utils.py:15-35."""
    st = _voxelize_issue(z, init_res, after_res)
    return _voxelize_finish(st, spf.read_counts([st['cnt']])[0])


KMAP_SPECS = [(3, 1)] + [(2, 2), (3, 1)] * 4       # the maps an SPVCNN-shaped encoder creates, in forward order


def _level_strides(specs):
    out, t = [], 1
    for _, stride in specs:
        if stride != 1:
            t *= stride
            out.append(t)
    return out


# U2MKD_GEOMETRY_BATCHED=0: the level-by-level form (one torch.unique per voxel set and per down-sampling: 6 host round
# trips per network) for A/B runs
_BATCHED = os.environ.get('U2MKD_GEOMETRY_BATCHED', '1') != '0'


def _prepare_geometry_level_by_level(x: SparseTensor, pres, vres):
    z = PointTensor(x.F, x.C.float())
    st = _voxelize_issue(z, pres, vres)
    x0 = _voxelize_finish(st, int(torch.unique(st['pc_hash']).shape[0]))
    with spf.deferred_range_check():          # the four down-samplings' out-of-range flag: one read instead of four
        spf.prefetch_kmaps(x0, KMAP_SPECS)
    return z, x0


def prepare_geometry_staged(items):
    """prepare_geometry_many as a GENERATOR that yields in front of each of its two host reads: a caller with other host work
    (train.KDStep: a step's forward and backward to queue) resumes it later and finds the counts already computed -- the same
    launches in the same order on whatever stream is current at each resumption, no waiting.  No context manager is held
    across a yield (a ``torch.no_grad()`` or stream context would leak into the caller's code): the CALLER sets them
    around every ``next()``.  Returns (StopIteration.value) what prepare_geometry_many returns."""
    if not _BATCHED:
        yield
        yield
        return [_prepare_geometry_level_by_level(*it[:3]) for it in items]
    tags = [it[3] if len(it) > 3 else None for it in items]      # (network tags: spf.prefetch_kmaps(tag=...))
    items = [it[:3] for it in items]
    states = [_voxelize_issue(PointTensor(x.F, x.C.float()), pres, vres) for x, pres, vres in items]
    # (the sizes are POSTED to the host mailbox by the slice that produces them and picked up by the next one: the host never
    # waits on a copy queued behind other streams' work, spf.post_counts)
    mail = spf.post_counts([st['cnt'] for st in states])
    yield
    sizes = spf.wait_counts(mail)                                              # round trip 1
    x0s = [_voxelize_finish(st, n) for st, n in zip(states, sizes)]
    totals = _level_strides(KMAP_SPECS)
    pyramids = [spf.DownsamplePyramid(x0.C, totals) if x0.C.shape[0] else None for x0 in x0s]
    mail = spf.post_counts([c for p in pyramids if p is not None for c in p.counts()])
    yield
    values = spf.wait_counts(mail)                                             # round trip 2
    out, at = [], 0
    for st, x0, pyr, tag in zip(states, x0s, pyramids, tags):
        if pyr is None:
            spf.prefetch_kmaps(x0, KMAP_SPECS, tag=tag)
        else:
            k = len(totals) + 1
            spf.prefetch_kmaps(x0, KMAP_SPECS, level_coords=pyr.finish(values[at:at + k]), tag=tag)
            at += k
        out.append((st['z'], x0))
    return out


def prepare_geometry_many(items):
    """prepare_geometry for several networks' inputs (``items`` = [(SparseTensor, pres, vres)]: the KD step's student and
    teacher) with TWO host round trips in total: the sizes of all stride-1 voxel sets are read together, then the sizes
    of every down-sampled level of every network (spf.DownsamplePyramid) together with the out-of-range flags -- where
    the one-network-at-a-time, one-level-at-a-time form stopped the host 6 times per network.  The work queued to the
    GPU and every result (voxel order, coordinates, kernel maps) are the same."""
    stages = prepare_geometry_staged(items)
    try:
        while True:
            next(stages)
    except StopIteration as done:
        return done.value


def prepare_geometry(x: SparseTensor, pres, vres):
    """The part of a forward pass that depends on the INPUT BATCH only and needs the host: points -> stride-1 voxels
    (``initial_voxelize``: the size of the voxel set) and the kernel maps of the whole encoder (``prefetch_kmaps``: the
    sizes of the four down-sampled levels) -- every host synchronisation of a training step is in here, two round trips
    (prepare_geometry_many).  Returns ``(z, x0)`` exactly as the first lines of the model's forward leave them; a
    forward that is handed the pair (``in_mod['_geometry']``) queues its launches without ever waiting for the GPU, so
    a trainer can prepare batch k+1 while step k's backward drains (train.KDStep ``prefetch=``)."""
    return prepare_geometry_many([(x, pres, vres)])[0]


def p2v_maps(coords, stride, z: PointTensor, plans=False):
    """(idx_query, counts) of the scatter-mean of ``z``'s points into the voxels ``coords`` of tensor stride ``stride``, cached
    in ``z.additional_features`` as core/models/utils.py:40-65 does.  ``plans``: also the scatter plan spvoxelize hangs on the
    index (a trainer preparing a batch ahead)."""
    cache = z.additional_features
    if cache is None or cache.get('idx_query') is None or cache['idx_query'].get(stride) is None:
        pc_hash = spf.sphash(_floor_coords(z.C, stride[0]))
        idx_query = spf.coords_table(coords).query_with_i32(pc_hash)
        counts = spf.spcount(spf._plan(idx_query, 'i32', lambda: idx_query.int().contiguous()), coords.shape[0])
        z.additional_features['idx_query'][stride] = idx_query
        z.additional_features['counts'][stride] = counts
    else:
        idx_query = cache['idx_query'][stride]
        counts = cache['counts'][stride]
    if plans:
        spf.voxelize_plan(idx_query, counts.shape[0])
    return idx_query, counts


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    """Scatter-mean point features into the voxels of ``x`` (utils.py:40-65)."""
    idx_query, counts = p2v_maps(x.C, x.s, z)
    inserted_feat = spf.spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(inserted_feat, x.C, x.s)
    new_tensor.cmaps = x.cmaps
    new_tensor.kmaps = x.kmaps
    return new_tensor


def v2p_maps(coords, stride, z: PointTensor, nearest=False, plans=False):
    """(idx_query [N, 8], weights [N, 8]) of the trilinear devoxelisation of the voxels ``coords`` (tensor stride
    ``stride``) at ``z``'s points, cached in ``z.idx_query`` / ``z.weights`` (utils.py:70-118).  ``plans``: also the backward's
    gather plan (spf.devoxelize_plan)."""
    if z.idx_query is None or z.weights is None or z.idx_query.get(stride) is None or z.weights.get(stride) is None:
        off = get_kernel_offsets(2, stride, 1, device=z.F.device)
        old_hash = spf.sphash(_floor_coords(z.C, stride[0]), off)          # [8, N]
        idx_kn = spf.coords_table(coords.to(z.F.device)).query(old_hash)
        weights, idx_query = spf.ti_weights_n8(z.C, idx_kn, scale=stride[0])   # [N,8], [N,8]
        if nearest:
            weights[:, 1:] = 0.
            idx_query[:, 1:] = -1
        z.idx_query[stride] = idx_query          # (the tensors voxel_to_point derives from z share these dictionaries)
        z.weights[stride] = weights
    else:
        idx_query, weights = z.idx_query.get(stride), z.weights.get(stride)
    if plans:
        spf.devoxelize_plan(idx_query, weights, coords.shape[0])
    return idx_query, weights


def voxel_to_point(x: SparseTensor, z: PointTensor, nearest=False) -> PointTensor:
    """Trilinear devoxelisation of ``x`` at the points of ``z`` (utils.py:70-118)."""
    idx_query, weights = v2p_maps(x.C, x.s, z, nearest)
    new_feat = spf.spdevoxelize(x.F, idx_query, weights)
    new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
    new_tensor.additional_features = z.additional_features
    return new_tensor


def fetch_idx(source_coords: torch.Tensor, target_coords: torch.Tensor) -> torch.Tensor:
    """Index of every source coordinate in ``target_coords`` (utils.py:121-135)."""
    assert isinstance(source_coords, torch.Tensor) and isinstance(target_coords, torch.Tensor)
    return spf.sphashquery(spf.sphash(source_coords), spf.sphash(target_coords))


class PointSyncBatchNorm1d(nn.SyncBatchNorm):
    """SyncBatchNorm over point features [N, C] (the converted nn.BatchNorm1d of the point branch)
    on the HIP SyncBatchNorm pieces (one small collective per pass); CPU tensors take torch's."""

    def forward(self, input):
        if input.is_cuda and input.dim() == 2:
            return spf.batch_norm(input, self)
        return super().forward(input)


class SparseSyncBatchNorm(nn.SyncBatchNorm):
    """SyncBatchNorm over SparseTensor features (utils.py:138-220): statistics over the voxels of
    ALL ranks; on the HIP device through spf.batch_norm (local slab statistics, one all_gather of
    [2C+1] floats, merge, fused normalise[+ReLU])."""

    def forward(self, input: SparseTensor) -> SparseTensor:
        if input.F.is_cuda:
            return fapply(input, spf.batch_norm, self)
        return fapply(input, super().forward)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        out = module
        if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
            from ..camera import BatchNorm2d as _CamBN, SyncBatchNorm2d
            klass = (SparseSyncBatchNorm if isinstance(module, spnn.BatchNorm)
                     else SyncBatchNorm2d if isinstance(module, (_CamBN, nn.BatchNorm2d)) else PointSyncBatchNorm1d)
            out = klass(module.num_features, module.eps, module.momentum, module.affine,
                        module.track_running_stats, process_group)
            if module.affine:
                with torch.no_grad():
                    out.weight = module.weight
                    out.bias = module.bias
            out.running_mean = module.running_mean
            out.running_var = module.running_var
            out.num_batches_tracked = module.num_batches_tracked
            if hasattr(module, 'qconfig'):
                out.qconfig = module.qconfig
        for name, child in module.named_children():
            out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        del module
        return out
