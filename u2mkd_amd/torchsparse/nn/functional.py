"""torchsparse.nn.functional on MI355X: every op is a HIP kernel behind the C ABI.

Call sites in the reference: core/models/utils.py:15-135 (sphash, sphashquery,
spcount, spvoxelize, spdevoxelize, calc_ti_weights) and every spnn.Conv3d of
core/models/build_blocks.py:25-80 (conv3d).  Semantics follow torchsparse
v1.4.0; the native data structure differs: a kernel map
is a neighbour table ``nbr[k][j]`` (plus the swapped-role table for strided
maps), consumed by an output-stationary MFMA kernel, instead of the
(nbmaps, nbsizes) rulebook driving gather -> GEMM -> scatter.  The rulebook is
still available lazily through the v1.4.0 ``kmap[0..2]`` interface.

GPU only: CPU tensors raise (there is no fallback path).
"""
from __future__ import annotations

import os
import threading

import weakref

import torch
from torch.autograd import Function

from ... import _host
from ... import _lib as L
from ..tensor import SparseTensor
from ..utils import make_ntuple
from .utils import get_kernel_offsets

_HOST = False          # lib/_u2mkd_host.so (C++ host side of the hottest operators) once loaded; None: U2MKD_HOST_OPS=0


def host_ops():
    global _HOST
    if _HOST is False:
        _HOST = _host.ops()
    return _HOST


__all__ = ['sphash', 'sphashquery', 'spcount', 'spvoxelize', 'spdevoxelize', 'calc_ti_weights',
           'spdownsample', 'conv3d', 'KernelMap', 'HashTable', 'ti_weights_n8', 'batch_norm']


def _i32(t):
    return t if t.dtype == torch.int32 else t.int()


# --------------------------------------------------------------------------- hash
def sphash(coords: torch.Tensor, offsets: torch.Tensor | None = None) -> torch.Tensor:
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    L.require_cuda(coords, offsets)
    coords = coords.contiguous()
    n = coords.shape[0]
    if offsets is None:
        out = torch.empty(n, dtype=torch.int64, device=coords.device)
        L.call('u2mkd_hash', L.ptr(coords), n, L.ptr(out), L.stream())
        return out
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    offsets = offsets.contiguous()
    k = offsets.shape[0]
    out = torch.empty(k, n, dtype=torch.int64, device=coords.device)
    L.call('u2mkd_kernel_hash', L.ptr(coords), L.ptr(offsets), n, k, L.ptr(out), L.stream())
    return out


class HashTable:
    """Device hash table over int64 keys -> index (smallest index on duplicates)."""

    def __init__(self, references: torch.Tensor):
        L.require_cuda(references)
        assert references.dtype == torch.int64
        references = references.contiguous().view(-1)
        self.n = references.shape[0]
        nbytes = L.load().u2mkd_hash_table_bytes(self.n)
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=references.device)
        L.call('u2mkd_hash_table_build', L.ptr(references), self.n, L.ptr(self.buf), L.stream())

    def query(self, queries: torch.Tensor) -> torch.Tensor:
        L.require_cuda(queries)
        assert queries.dtype == torch.int64
        q = queries.contiguous()
        out = torch.empty_like(q)
        L.call('u2mkd_hash_table_query', L.ptr(self.buf), self.n, L.ptr(q), q.numel(), L.ptr(out), L.stream())
        return out

    def query_with_i32(self, queries: torch.Tensor) -> torch.Tensor:
        """``query`` whose result carries its int32 copy (the form the voxelise / count kernels read) as the cached 'i32'
        plan of the returned tensor: one launch instead of the query and a conversion per consumer."""
        L.require_cuda(queries)
        assert queries.dtype == torch.int64
        q = queries.contiguous()
        out = torch.empty_like(q)
        out32 = torch.empty(q.shape, dtype=torch.int32, device=q.device)
        L.call('u2mkd_hash_table_query2', L.ptr(self.buf), self.n, L.ptr(q), q.numel(), L.ptr(out), L.ptr(out32), L.stream())
        out.__dict__.setdefault('_u2mkd_plans', {})['i32'] = ((out._version,), out32)
        return out


def coords_table(coords: torch.Tensor) -> HashTable:
    """The hash table over a coordinate set's hashes, built ONCE per coordinate tensor (cached on it with its version counter,
    ``_plan``): a level's coordinates are hashed and inserted by build_kmap (twice per level: the 3 x 3 x 3 and the 2 x 2 x 2
    map), point_to_voxel and voxel_to_point alike -- 44 builds (4 launches each) per KD step, 10 distinct tables."""
    c = _i32(coords).contiguous()
    return _plan(c, 'hashtable', lambda: HashTable(sphash(c)))


def sphashquery(queries: torch.Tensor, references: torch.Tensor) -> torch.Tensor:
    """Index of every query hash in ``references`` (-1 on miss), query shape kept."""
    return HashTable(references).query(queries)


def spcount(coords: torch.Tensor, num: int) -> torch.Tensor:
    L.require_cuda(coords)
    idx = _i32(coords).contiguous()
    out = torch.zeros(int(num), dtype=torch.int32, device=idx.device)
    L.call('u2mkd_count', L.ptr(idx), idx.numel(), L.ptr(out), int(num), L.stream())
    return out


# ------------------------------------------------------------ bf16 storage
_BF16_ROWS = os.environ.get('U2MKD_BF16_ROWS', '1') != '0'      # 0: autocast keeps fp32 rows between the sparse operators


def bf16_rows() -> bool:
    """True when the sparse operators keep their feature rows in bf16: under ``torch.autocast('cuda', bfloat16)``.
    torchsparse v1.4.0 decorates its conv / voxelize / devoxelize functions ``custom_fwd(cast_inputs=torch.half)``
    (SURVEY.md Appendix A-6), so under the reference's amp (core/nusc_trainers.py:285) rows travel in half between
    them and nn.BatchNorm1d passes half through; here the reduced type is bf16 (BASELINE.json configs[4]), every
    accumulation, statistic and weight gradient stays fp32.  fp16 autocast keeps fp32 rows (no fp16 kernels)."""
    return _BF16_ROWS and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


def _rows(t, b16):
    """contiguous rows in the storage type of the call"""
    return t.contiguous().to(torch.bfloat16 if b16 else torch.float32)


# U2MKD_PREFETCH_PLANS=0 (default): every derived index structure (tile / pair schedules, weight-gradient pairs, the scatter plans
# of voxelize / devoxelize, window plans, point <-> pixel plans) is built at its first use inside the forward / backward pass, as in
# rounds 1-5; 1: a trainer that prepares the next batch's geometry ahead builds them there too (train.KDStep, kd.py; 2: and the
# teacher's schedules).  Round 6 built it to take ~350 small launches per step off the student's stream -- and measured the step
# 0.4-0.5 ms SLOWER (2: 1.5 ms; with the geometry stream on a hardware queue of its own 2.4 ms): on the runtime's four hardware
# queues the geometry stream shares the main stream's queue, so the launches leave the forward but not the queue they wait in, and
# wherever they do run next to the step's kernels they cost more than in line (NOTES N10.9).  Kept as a switch, results identical.
_PREFETCH_PLANS = os.environ.get('U2MKD_PREFETCH_PLANS', '0') != '0'
_PREFETCH_PARTS = os.environ.get('U2MKD_PREFETCH_PARTS', 'ABCD')      # (A/B runs: A maps, B kernel-map schedules, C window plans, D fusion plans)


def prefetch_plans_enabled():
    return _PREFETCH_PLANS


# ------------------------------------------------- destination-sorted scatter plans
def _csr_by_destination(keys: torch.Tensor, nv: int):
    """(entry order int32 [E], segment offsets int32 [nv+1]) of the entries with 0 <= key < nv, grouped by key, ascending
    entry id inside a group (what a stable argsort by key gives; the entries past seg[nv] are 0 and never addressed).
    One call, no host sync (csrc/csr.hip: counting sort)."""
    k32 = _i32(keys).contiguous().view(-1)
    e = k32.numel()
    order = torch.empty(e, dtype=torch.int32, device=keys.device)
    seg = torch.empty(nv + 1, dtype=torch.int32, device=keys.device)
    ws = torch.empty(max(L.load().u2mkd_csr_workspace_bytes(e, nv), 16), dtype=torch.uint8, device=keys.device)
    L.call('u2mkd_csr_build', L.ptr(k32), e, nv, L.ptr(ws), L.ptr(order), L.ptr(seg), L.stream())
    return order, seg


def _plan(t: torch.Tensor, name: str, build, *deps):
    """Cache a derived index structure on the index tensor it was derived from (the point<->voxel
    index tensors are themselves cached by the model in z.idx_query / z.additional_features).  The entry is
    rebuilt when the index tensor, or any tensor in ``deps`` that is baked into the plan, has been written in
    place or replaced since (torch version counters + data pointers): torchsparse's functions carry no hidden
    state, so a caller may edit ``idx`` / ``weights`` between two calls."""
    cache = t.__dict__.setdefault('_u2mkd_plans', {})
    stamp = (t._version,) + tuple((d.data_ptr(), d._version) for d in deps)
    hit = cache.get(name)
    if hit is None or hit[0] != stamp:
        hit = cache[name] = (stamp, build())
    return hit[1]


def _segment_sum(src, erow, ew, seg, nv, mean):
    """rows of src (fp32 or bf16: the output has the same type) summed per destination segment"""
    c = src.shape[1]
    out = torch.empty(nv, c, dtype=src.dtype, device=src.device)
    L.call('u2mkd_segment_sum_bf16' if src.dtype == torch.bfloat16 else 'u2mkd_segment_sum', L.ptr(src), c, L.ptr(erow),
           L.ptr(ew), L.ptr(seg), nv, int(mean), L.ptr(out), L.stream())
    return out


# ----------------------------------------------------------------- voxelize
class VoxelizeFunction(Function):
    @staticmethod
    def forward(ctx, feats, coords, counts):
        L.require_cuda(feats, coords, counts)
        b16 = bf16_rows() and feats.shape[1] % 4 == 0 and feats.shape[1] >= 16    # (coordinate means stay fp32: c = 4)
        feats = _rows(feats, b16)
        coords = _i32(coords).contiguous()
        counts = _i32(counts).contiguous()
        n, c = feats.shape
        nv = counts.shape[0]
        if n == 0:
            ctx.for_backwards = (coords, counts, n)
            return torch.zeros(nv, c, dtype=feats.dtype, device=feats.device)
        if c % 4 == 0:
            # deterministic scatter-mean: points grouped by voxel once per map, then a gather-sum
            order, seg = voxelize_plan(coords, nv)
            out = _segment_sum(feats, order, None, seg, nv, True)
        else:
            out = torch.zeros(nv, c, dtype=torch.float32, device=feats.device)
            L.call('u2mkd_voxelize_forward', L.ptr(feats), L.ptr(coords), L.ptr(counts), n, nv, c, L.ptr(out),
                   L.stream())
        ctx.for_backwards = (coords, counts, n)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        coords, counts, n = ctx.for_backwards
        b16 = grad_output.dtype == torch.bfloat16 and grad_output.shape[1] % 4 == 0
        g = _rows(grad_output, b16)
        nv, c = g.shape
        gi = torch.empty(n, c, dtype=g.dtype, device=g.device)
        L.call('u2mkd_voxelize_backward_bf16' if b16 else 'u2mkd_voxelize_backward', L.ptr(g), L.ptr(coords),
               L.ptr(counts), n, nv, c, L.ptr(gi), L.stream())
        return gi, None, None


def _idx32(coords):
    # the model caches idx_query (int64, from sphashquery) per stride; keep its int32 form -- and the
    # destination-sorted plan hung on it -- alive on that cached tensor instead of re-deriving per call
    if coords.dtype != torch.int32 or not coords.is_contiguous():
        coords = _plan(coords, 'i32', lambda: coords.int().contiguous())
    return coords


def voxelize_plan(coords, nv):
    """(order, seg) of spvoxelize's scatter-mean: the points grouped by voxel (cached on the index tensor).  ``coords`` = the
    point -> voxel index as spvoxelize receives it."""
    coords = _idx32(coords)
    return _plan(coords, 'vox_csr_%d' % nv, lambda: _csr_by_destination(coords, nv))


def spvoxelize(feats, coords, counts):
    return VoxelizeFunction.apply(feats, _idx32(coords), counts)


# --------------------------------------------------------------- devoxelize
class DevoxelizeFunction(Function):
    @staticmethod
    def forward(ctx, feats, coords, weights):
        L.require_cuda(feats, coords, weights)
        b16 = bf16_rows() and feats.shape[1] % 4 == 0 and feats.shape[1] >= 16
        feats = _rows(feats, b16)
        coords = _i32(coords).contiguous()
        weights = weights.contiguous().float()
        nv, c = feats.shape
        n = coords.shape[0]
        assert coords.shape == (n, 8) and weights.shape == (n, 8), (coords.shape, weights.shape)
        out = torch.empty(n, c, dtype=feats.dtype, device=feats.device)
        L.call('u2mkd_devoxelize_forward_bf16' if b16 else 'u2mkd_devoxelize_forward', L.ptr(feats), L.ptr(coords),
               L.ptr(weights), n, c, L.ptr(out), L.stream())
        ctx.for_backwards = (coords, weights, nv)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        coords, weights, nv = ctx.for_backwards
        g = _rows(grad_output, grad_output.dtype == torch.bfloat16 and grad_output.shape[1] % 4 == 0)
        n, c = g.shape
        if n == 0:
            return torch.zeros(nv, c, dtype=g.dtype, device=g.device), None, None
        if c % 4 == 0:
            erow, ew, seg = devoxelize_plan(coords, weights, nv)
            gi = _segment_sum(g, erow, ew, seg, nv, False)
        else:
            g = g.float()
            gi = torch.zeros(nv, c, dtype=torch.float32, device=g.device)
            L.call('u2mkd_devoxelize_backward', L.ptr(g), L.ptr(coords), L.ptr(weights), n, nv, c, L.ptr(gi),
                   L.stream())
        return gi, None, None


def devoxelize_plan(coords, weights, nv):
    """(entry row, entry weight, seg) of spdevoxelize's BACKWARD (a scatter of 8 weighted corners per point, as a gather-sum per
    voxel): zero-weight corners dropped, the rest grouped by voxel -- keys + counting sort + (row, weight) in one call.  Cached on
    the index tensor; ``coords`` int32 [n, 8] contiguous, ``weights`` f32 [n, 8] contiguous, as DevoxelizeFunction holds them."""
    n = coords.shape[0]

    def build():
        dev = coords.device
        erow = torch.empty(8 * n, dtype=torch.int32, device=dev)
        ew = torch.empty(8 * n, dtype=torch.float32, device=dev)
        seg = torch.empty(nv + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(L.load().u2mkd_devoxelize_plan_workspace_bytes(n, nv), 16), dtype=torch.uint8, device=dev)
        L.call('u2mkd_devoxelize_plan', L.ptr(coords), L.ptr(weights), n, nv, L.ptr(ws), L.ptr(erow), L.ptr(ew), L.ptr(seg),
               L.stream())
        return erow, ew, seg
    return _plan(coords, 'devox_csr_%d' % nv, build, weights)


def spdevoxelize(feats, coords, weights):
    return DevoxelizeFunction.apply(feats, coords, weights)


def ti_weights_n8(coords: torch.Tensor, idx_query_kn: torch.Tensor, scale=1):
    """Fused F.calc_ti_weights + the two [8,N]->[N,8] transposes of
    core/models/utils.py:94-95.  Returns (weights f32 [N,8], idx int32 [N,8])."""
    L.require_cuda(coords, idx_query_kn)
    coords = coords.contiguous().float()
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    idx = idx_query_kn.contiguous()
    assert idx.dtype == torch.int64 and idx.shape == (8, coords.shape[0]), (idx.dtype, idx.shape)
    n = coords.shape[0]
    w = torch.empty(n, 8, dtype=torch.float32, device=coords.device)
    i8 = torch.empty(n, 8, dtype=torch.int32, device=coords.device)
    L.call('u2mkd_ti_weights', L.ptr(coords), L.ptr(idx), n, float(scale), L.ptr(w), L.ptr(i8), L.stream())
    return w, i8


def calc_ti_weights(coords, idx_query, scale=1):
    """v1.4.0 signature: returns [8,N] (a transposed view of the fused [N,8]
    result, so the caller's ``.transpose(0,1).contiguous()`` is free)."""
    with torch.no_grad():
        w, _ = ti_weights_n8(coords, idx_query, scale)
        return w.t()


# --------------------------------------------------------------- downsample
def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    """Output coordinates of a strided conv, sorted by (b,x,y,z) (Appendix A-4).
    Only the stride[k] in {1, kernel_size[k]} branch is on the U2MKD path."""
    L.require_cuda(coords)
    stride = make_ntuple(stride, ndim=3)
    kernel_size = make_ntuple(kernel_size, ndim=3)
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    if not all(stride[k] in (1, kernel_size[k]) for k in range(3)):
        raise NotImplementedError('spdownsample: only stride in {1, kernel_size} is supported '
                                  '(the only form U2MKD uses: k=2, s=2)')
    ss = [stride[k] * tensor_stride[k] for k in range(3)]
    coords = _i32(coords).contiguous()
    n = coords.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=coords.device)
    flag = _range_flag(coords.device)
    L.call('u2mkd_downsample_keys_checked', L.ptr(coords), n, ss[0], ss[1], ss[2], L.ptr(keys), L.ptr(flag), L.stream())
    uniq = torch.unique(keys)  # sorted int64 == (b,x,y,z) lexicographic
    # (the unique above has just synchronised the stream: reading the 4-byte flag costs no queue drain -- but it is a
    # second round trip per level; a caller that builds several levels in a row reads the flag once, at the end)
    if _deferred()[0]:
        _deferred().append(flag)
    elif n:
        _check_range_flag(flag)
    out = torch.empty(uniq.shape[0], 4, dtype=torch.int32, device=coords.device)
    L.call('u2mkd_unpack_keys', L.ptr(uniq), uniq.shape[0], L.ptr(out), L.stream())
    return out


def unique_sorted_deferred(keys):
    """The sorted unique values of an int64 vector WITHOUT reading their number back (``torch.unique`` stops the host for
    it): returns (buffer [n] whose first ``count`` entries are the unique values in ascending order, ``count`` as a 0-d
    int64 device tensor).  A caller that needs several such sets issues them all and reads the counts in ONE round trip
    (:func:`read_counts`)."""
    n = keys.shape[0]
    srt = torch.sort(keys).values
    head = torch.ones(n, dtype=torch.bool, device=keys.device)
    if n > 1:
        torch.ne(srt[1:], srt[:-1], out=head[1:])
    pos = torch.cumsum(head, 0)
    buf = torch.empty_like(srt)
    buf.scatter_(0, pos - 1, srt)                      # equal keys write the same value to the same slot
    return buf, (pos[-1] if n else torch.zeros((), dtype=torch.int64, device=keys.device))


# ---- host mailbox for the sizes the host has to know (csrc/mailbox.hip) -----------------------------------------------------
# U2MKD_COUNTS_MAILBOX=0: every read is a stream-ordered device-to-host copy + stream synchronisation (rounds 1-5; A/B runs)
_MAILBOX_ON = os.environ.get('U2MKD_COUNTS_MAILBOX', '1') != '0'
_MAIL_SLOTS, _MAIL_WORDS = 64, 32          # a ring of slots of 32 words, one per value: (sequence number << 32) | value
_MAIL = {}                                 # 'mem': numpy view of the ring, 'base': its address, 'seq': last sequence number


def _mailbox():
    if not _MAIL:
        import ctypes
        import numpy as np
        nbytes = _MAIL_SLOTS * _MAIL_WORDS * 8
        base = L.load().u2mkd_mailbox_alloc(nbytes)
        if not base:
            raise RuntimeError('u2mkd_mailbox_alloc failed: ' + L.load().u2mkd_last_error().decode('utf-8', 'replace'))
        _MAIL['base'] = base
        _MAIL['mem'] = np.ctypeslib.as_array((ctypes.c_uint64 * (_MAIL_SLOTS * _MAIL_WORDS)).from_address(base))
        _MAIL['seq'] = 0
        _MAIL['ptrs'] = (ctypes.c_void_p * 32)()
    return _MAIL


class _PostedCounts:
    """Handle of :func:`post_counts`: the values are on their way to the host."""
    __slots__ = ('n', 'at', 'seq', 'tensors', 'stream', 'values')


def post_counts(tensors):
    """ISSUE side of a host read of several 0-d / 1-element device integers: one small kernel on the current stream stores them
    into mapped host memory (u2mkd_mailbox_post) -- queue it right behind their producers; :func:`wait_counts` later polls for
    them.  Unlike a stream-ordered copy issued at the time of the read, nothing here waits for what OTHER streams that share
    the hardware queue have queued in between (NOTES N10)."""
    h = _PostedCounts()
    h.tensors = list(tensors)
    h.n = len(h.tensors)
    h.values = None
    if h.n == 0:
        h.values = []
        return h
    if not (_MAILBOX_ON and h.n <= 32 and all(t.is_cuda and t.numel() == 1 and t.dtype in (torch.int64, torch.int32) for t in h.tensors)):
        h.seq = None                  # (CPU tensors, the A/B switch: the stream-ordered copy at wait time)
        return h
    m = _mailbox()
    m['seq'] += 1
    h.seq = m['seq']
    h.at = (h.seq % _MAIL_SLOTS) * _MAIL_WORDS
    is64 = 0
    for i, t in enumerate(h.tensors):
        m['ptrs'][i] = t.data_ptr()
        if t.dtype == torch.int64:
            is64 |= 1 << i
    h.stream = torch.cuda.current_stream()
    import ctypes
    L.call('u2mkd_mailbox_post', ctypes.addressof(m['ptrs']), is64, h.n, m['base'] + h.at * 8, h.seq, L.stream())
    return h


def wait_counts(h, timeout_s=60.0):
    """HOST side: the values of :func:`post_counts` as Python ints (polls the mailbox's sequence word)."""
    if h.values is not None:
        return h.values
    if h.seq is None:
        h.values = torch.stack([t.reshape(()).to(torch.int64) for t in h.tensors]).tolist()
        return h.values
    # word i = (seq << 32) | value: ready when its upper half is this post's sequence number (csrc/mailbox.hip)
    words = _MAIL['mem'][h.at:h.at + h.n]
    tag = h.seq & 0xffffffff
    if not bool(((words >> 32) == tag).all()):
        import time
        t0, spins = time.perf_counter(), 0
        while not bool(((words >> 32) == tag).all()):
            spins += 1
            if spins & 255 == 0:
                if time.perf_counter() - t0 > timeout_s:
                    h.stream.synchronize()          # (raises if the producing stream faulted)
                    if not bool(((words >> 32) == tag).all()):
                        raise RuntimeError('wait_counts: the mailbox was not written (sequence %d)' % h.seq)
                time.sleep(0)
    h.values = [int(w) & 0xffffffff for w in words]
    if 0xffffffff in h.values:
        raise OverflowError('wait_counts: a posted value is outside [0, 2^32 - 2]')
    h.tensors = None
    return h.values


def read_counts(tensors):
    """One host round trip for a list of 0-d / 1-element device integers (issue + wait; a caller with other host work in
    between calls :func:`post_counts` / :func:`wait_counts` itself)."""
    if not tensors:
        return []
    return wait_counts(post_counts(tensors))


class DownsamplePyramid:
    """Output coordinates of a CHAIN of k = 2, s = 2 down-samplings (the encoder's four levels) from the stride-1
    coordinates, with one host round trip for all of them instead of one ``torch.unique`` per level: level l of the
    chain is unique(floor(c / 2^l) * 2^l) of the level below, which equals the same expression applied to the base
    coordinates, so every level's key set is issued from the base (keys -> sort -> deferred unique) and the sizes are
    read together with the out-of-range flag.  ``issue`` queues the kernels, ``counts`` lists the device integers the
    caller reads (read_counts), ``finish`` takes their values and returns {total stride: coords [n_l, 4] int32} in the
    order spdownsample leaves them (sorted by (b, x, y, z))."""

    def __init__(self, coords, totals):
        L.require_cuda(coords)
        self.coords = _i32(coords).contiguous()
        self.totals = [make_ntuple(t, ndim=3) for t in totals]
        n = self.coords.shape[0]
        self.flag = _range_flag(self.coords.device)
        self.sets = []
        for ss in self.totals:
            keys = torch.empty(n, dtype=torch.int64, device=self.coords.device)
            L.call('u2mkd_downsample_keys_checked', L.ptr(self.coords), n, ss[0], ss[1], ss[2], L.ptr(keys), L.ptr(self.flag), L.stream())
            self.sets.append(unique_sorted_deferred(keys))

    def counts(self):
        return [c for _, c in self.sets] + [self.flag]

    def finish(self, values):
        assert len(values) == len(self.sets) + 1
        if values[-1] != 0:
            self.flag.zero_()
            raise ValueError(_RANGE_MESSAGE)
        out = {}
        for ss, (buf, _), cnt in zip(self.totals, self.sets, values):
            c = torch.empty(cnt, 4, dtype=torch.int32, device=buf.device)
            L.call('u2mkd_unpack_keys', L.ptr(buf), cnt, L.ptr(c), L.stream())
            out[ss] = c
        return out


_RANGE_MESSAGE = ('spdownsample: coordinates outside the packed key range (|x|, |y|, |z| < 131072 voxels, '
                  '0 <= batch index < 512)')
_RANGE_FLAGS = {}
_DEFERRED_RANGE_CHECK = threading.local()      # .state = [active, flag, flag, ...], one per host thread


def _deferred():
    st = getattr(_DEFERRED_RANGE_CHECK, 'state', None)
    if st is None:
        st = _DEFERRED_RANGE_CHECK.state = [False]
    return st


def _check_range_flag(flag):
    if int(flag) != 0:
        flag.zero_()
        raise ValueError(_RANGE_MESSAGE)


class deferred_range_check:
    """spdownsample calls inside this context leave their out-of-range flag unread; the flag (sticky on the device) is
    read ONCE when the context ends -- one host round trip for a whole pyramid of down-samplings instead of one per
    level.  The error is the same, raised a few launches later."""

    def __enter__(self):
        st = _deferred()
        self._outer = st[0]
        st[0] = True
        self._start = len(st)
        return self

    def __exit__(self, exc_type, exc, tb):
        st = _deferred()
        flags = st[self._start:]
        del st[self._start:]
        st[0] = self._outer
        if exc_type is None and not self._outer:
            seen = set()
            for f in flags:
                if id(f) not in seen:
                    seen.add(id(f))
                    _check_range_flag(f)
        elif self._outer:
            st.extend(flags)
        return False


def _range_flag(device):
    """int32 the key kernel raises when a coordinate cannot be packed (see u2mkd_downsample_keys_checked).  One per
    (device, stream), like ``_scratch``: the frozen teacher on its side stream and the student on the main stream call
    ``spdownsample`` concurrently, and a shared flag could be read or cleared by the wrong caller."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    f = _RANGE_FLAGS.get(key)
    if f is None:
        f = _RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


# -------------------------------------------------------------- kernel maps
def _pairs_mode(cin, cout, n_rows=0):
    """Schedule choice, measured on MI355X over the SPVCNN layer shapes (tools/ab_schedule.py):
    the pair schedule wins from 96x96 channels up (2x at 256x256), the tile schedule below
    (the pair schedule's extra round trip of P x cout floats costs more than it saves).  At the boundary itself (64 x 128,
    128 x 64) the tile schedule wins on the large voxel sets (80k rows: 58-66 us against 71-88 us) and loses on the small ones
    (16k rows: 37-43 us against 34 us)."""
    env = os.environ.get('U2MKD_CONV_SCHEDULE')
    if env in ('tiles', 'pairs'):
        return env == 'pairs'
    if cout % 4:
        return False
    return cin * cout > 8192 or (cin * cout == 8192 and n_rows < 24000)


# wide layers on the bf16x3 pair kernel (csrc/conv_px3.hip); U2MKD_PAIRS_X3=0 keeps them on the f32-MFMA pair kernel
_PAIRS_X3 = os.environ.get('U2MKD_PAIRS_X3', '1') != '0'

_SCRATCH = {}


def _scratch(nbytes, device):
    """Grow-only per-device scratch (the pair schedule's y rows).  It is produced and consumed by
    two launches that follow each other on one stream, so one buffer serves every layer."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


_TILE_SPLIT = tuple(int(v) for v in os.environ.get('U2MKD_TILE_SPLIT', '30,60').split(','))   # blocks above which a 64-row tile
# is processed as 2 / 4 work items (measured: tools/ab_tp.py)


class TileSchedule:
    """Output-stationary walk of a neighbour table [K, N]: 64-row tiles of the table in SORTED
    row order, each tile visits the union of its rows' offsets serially.  Rows are sorted by
    their 27-bit neighbour mask (tiles of one mask skip empty (block, offset) slots), and the
    tiles are LAUNCHED heaviest-first: tiles made of rare masks walk up to 27 offsets with ~2
    useful rows each and are the kernel's critical path (in-kernel timestamps, DESIGN.md
    section 6), so they must not be the last to start.  Built without host synchronisation.
        nbr_s int32 [K, N], order int32 [N] (original row of sorted row),
        tile_order int32 [ceil(N / 64)] (tile ids by descending offset count)"""

    def __init__(self, tbl):
        self.k, self.n = tbl.shape
        k, n = tbl.shape
        dev = tbl.device
        mask = torch.empty(n, dtype=torch.int32, device=dev)
        if n:
            L.call('u2mkd_kmap_rowmask', L.ptr(tbl), n, k, L.ptr(mask), L.stream())
        order = torch.argsort(mask, stable=True)
        self.nbr_s = tbl.index_select(1, order).contiguous()
        self.order = order.int()
        t = (n + 63) // 64
        # Work of a set of rows = its 16-pair MFMA blocks, sum over offsets of ceil(pairs / 16).  tile_order = the
        # 64-row tiles by descending blocks (offset-walking kernels).  Work items of the tile-pair kernel
        # (u2mkd_conv_forward_tiles): a tile is a SERIAL chain of blocks, so the heaviest tiles (rows with rare
        # neighbour masks: up to 62 blocks against a mean of 16) are cut into halves / quarters; items are listed
        # heaviest first, item = tile << 4 | sub << 2 | lg.  Built on the device by two launches
        # (csrc/schedule.hip: block counts per tile / half / quarter from ballots over the sorted masks, then one
        # workgroup's stable counting sorts) -- the torch formulation it replaces (bit-plane sums, three argsorts,
        # where / cat) was ~35 launches per schedule, a fifth of a training step's launches.  No host sync: the
        # item count stays on the device.
        self.tile_order = torch.empty(t, dtype=torch.int32, device=dev)
        self.items = torch.empty(7 * t, dtype=torch.int32, device=dev)
        self.n_items = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(t, 1) * 8, dtype=torch.int32, device=dev)
        L.call('u2mkd_tile_schedule', L.ptr(mask), L.ptr(self.order), n, k, _TILE_SPLIT[0], _TILE_SPLIT[1], L.ptr(ws),
               L.ptr(self.tile_order), L.ptr(self.items), L.ptr(self.n_items), L.stream())

    def tiles(self):
        return self.nbr_s, self.order

    def run(self, feats, weight, transpose, cout, kflip, out, epilogue=None):
        """out[j] = sum_k feats[tbl[k][j]] @ B_k with B_k[col][ci] = weight[kk][ci][col] (transpose, the forward) or
        weight[kk][col][ci] (the input gradient), kk = K-1-k if kflip else k; weight = `kernel` [K, cin, cout].
        ``epilogue`` = (scale [cout], shift [cout], residual [n, cout] or None, relu): a folded eval-mode BatchNorm
        (conv_eval_affine); only where ``supports_epilogue`` says so."""
        n_in, cin = feats.shape
        # fp32 rows: f16x2 (arith 4) where the tile kernel has it, else the library default (bf16x3 / f32)
        ar = 4 if feats.dtype == torch.float32 and _tiles_arith(cin, cout, self.k) == 4 else 0
        if epilogue is not None:
            sc, sh, res, relu = epilogue
            wf = _weight_layout(weight, transpose, True, arith=ar)
            L.call('u2mkd_conv_forward_tiles_ep', L.ptr(feats), n_in, cin, L.ptr(wf), cout, L.ptr(self.nbr_s),
                   L.ptr(self.order), L.ptr(self.items), L.ptr(self.n_items), self.n, self.k, int(kflip), ar, L.ptr(sc), L.ptr(sh),
                   L.ptr(res), int(relu), L.ptr(out), L.stream())
            return out
        if feats.dtype == torch.bfloat16:        # bf16 storage: rows in and out bf16, one bf16 weight plane
            wf = _weight_layout(weight, transpose, True, arith=3)
            L.call('u2mkd_conv_forward_tiles_bf16', L.ptr(feats), n_in, cin, L.ptr(wf), cout, L.ptr(self.nbr_s),
                   L.ptr(self.order), L.ptr(self.items), L.ptr(self.n_items), self.n, self.k, int(kflip), L.ptr(out),
                   L.stream())
        elif L.load().u2mkd_conv_tiles_supported(cin, cout, self.k):
            wf = _weight_layout(weight, transpose, True, arith=ar)
            L.call('u2mkd_conv_forward_tiles', L.ptr(feats), n_in, cin, L.ptr(wf), cout, L.ptr(self.nbr_s),
                   L.ptr(self.order), L.ptr(self.items), L.ptr(self.n_items), self.n, self.k, int(kflip), ar, L.ptr(out),
                   L.stream())
        else:
            wt = _weight_layout(weight, transpose, False)
            L.call('u2mkd_conv_forward_sorted', L.ptr(feats), n_in, cin, L.ptr(wt), cout, L.ptr(self.nbr_s),
                   L.ptr(self.order), L.ptr(self.tile_order), self.n, self.k, int(kflip), L.ptr(out), L.stream())
        return out


class BnStats:
    """What a convolution's store can hand the train-mode BatchNorm that follows it: slab partials of its output (``partial``
    [slabs][2][C], slabs of ``slab_rows`` rows) and which tensor they describe (``of`` = (address, rows, channels))."""
    __slots__ = ('partial', 'slab_rows', 'of')

    def __init__(self):
        self.partial = self.slab_rows = self.of = None

    def describes(self, x):
        return self.partial is not None and self.of == (x.data_ptr(), x.shape[0], x.shape[1]) and x.dtype == torch.float32


# the sink a caller opens around ONE convolution whose output goes straight into a train-mode BatchNorm (lidar.blocks.
# FusedSequential); a store that can take the statistics fills it (PairSchedule.run), anything else leaves it empty and the
# BatchNorm runs its own statistics pass.  U2MKD_CONV_BN_STATS=1 opens it; default 0: built, verified
# (tests/test_gpu_torchsparse_ops.py::test_conv_store_takes_the_batch_norm_statistics), and measured without gain -- 29 of the
# KD student's BatchNorms lose their statistics launch, the step stays within +-0.15 ms: the gather-sum that owns 32-row slabs
# is 4-18 us slower than the flat one and the merge reads four times the partials (tools/exp_gather_sum_stats.py, NOTES N10.14).
BN_STATS_SINK = [None]
_CONV_BN_STATS = os.environ.get('U2MKD_CONV_BN_STATS', '0') != '0'


def conv_bn_stats_enabled():
    return _CONV_BN_STATS


# where the gather-sum with statistics + the BatchNorm from partials beat gather-sum + the BatchNorm's own three launches
# (tools/exp_gather_sum_stats.py, MI355X): a workgroup of the statistics form owns 32 rows, so on long, narrow outputs it has a
# third of the plain form's workgroups in flight -- 80 000 x 96: 54 -> 69 us the pair, 32 000 x 256: 63 -> 60, 16 000 x 256: 47 -> 45
_STATS_MAX_ROWS, _STATS_MIN_COUT = (int(v) for v in os.environ.get('U2MKD_CONV_BN_STATS_SHAPES', '100000000,8').split(','))


def _stats_in_store(n_rows, cout):
    return n_rows <= _STATS_MAX_ROWS and cout >= _STATS_MIN_COUT and bool(L.load().u2mkd_pairs_gather_sum_stats_supported(cout))


class PairSchedule:
    """Offset-grouped pair list of a kernel map (u2mkd_pairs_build): every (input i, output j)
    pair, grouped by offset and padded to 128 entries per offset (two 64-pair tiles).  A conv is two launches: one
    dense MFMA stage per 64-pair tile into a scratch y (all tiles independent -- no serial
    walk), then a gather-sum over each row's <= K slots in ascending offset order
    (deterministic, no atomics).  One schedule serves the forward (gather inputs, sum per
    output), the transposed conv and the input gradient (gather outputs, sum per input).
    Everything is sized by the capacity K * min(n_in, n_out); the host never reads the
    pair count."""

    def __init__(self, nbr, n_in):
        k, n_out = nbr.shape
        dev = nbr.device
        lib = L.load()
        self.k, self.n_in, self.n_out = k, n_in, n_out
        self.cap = int(lib.u2mkd_pairs_capacity(n_in, n_out, k))
        nblocks = max((n_out + 1023) // 1024, 1)
        zeros = torch.zeros(k + 2, dtype=torch.int32, device=dev)      # (one fill for both zero-initialised pieces)
        nbsizes = zeros[:k]
        block_counts = torch.empty(k, nblocks, dtype=torch.int32, device=dev)
        self.pair_in = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.pair_out = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.pos_out = torch.empty(n_out, k, dtype=torch.int32, device=dev)
        self.pos_in = torch.full((n_in, k), -1, dtype=torch.int32, device=dev)
        self.tile_k = torch.empty(self.cap // 64, dtype=torch.int32, device=dev)
        self.meta = zeros[k:]
        if n_out and n_in:
            st = L.stream()
            L.call('u2mkd_kmap_sizes', L.ptr(nbr), n_out, k, L.ptr(nbsizes), L.ptr(block_counts), st)
            L.call('u2mkd_pairs_build', L.ptr(nbr), n_out, n_in, k, L.ptr(nbsizes), L.ptr(block_counts),
                   L.ptr(self.pair_in), L.ptr(self.pair_out), L.ptr(self.pos_out), L.ptr(self.pos_in),
                   L.ptr(self.tile_k), L.ptr(self.meta), st)
        self.nbsizes = nbsizes

    def run(self, feats, wt, cout, swap, out, variant=0, fragments=False, epilogue=None):
        """swap = False: out[j] = sum_k feats[in_k(j)] @ B_k  (rows of out = the map's outputs)
        swap = True:  out[i] = sum_k feats[out_k(i)] @ B_k (rows of out = the map's inputs);
        B_k = wt[k] as [cout][cin], or (fragments) the arith-2 fragment layout of it for the bf16x3 kernel (fragments = 2: the
        arith-4 layout, the same kernel in f16x2 arithmetic)."""
        n, cin = feats.shape
        st = L.stream()
        idx, pos, n_rows = (self.pair_out, self.pos_in, self.n_in) if swap else (self.pair_in, self.pos_out, self.n_out)
        if n_rows == 0:
            return out
        y = _scratch(self.cap * cout * 4, feats.device)
        if feats.dtype == torch.bfloat16:        # bf16 storage (wt = the arith-3 fragments): scratch rows y in bf16 too
            L.call('u2mkd_conv_forward_pairs_bf16', L.ptr(feats), n, cin, L.ptr(wt), cout, L.ptr(idx), L.ptr(self.tile_k),
                   L.ptr(self.meta), self.cap, self.k, L.ptr(y), st)
            L.call('u2mkd_pairs_gather_sum_bf16', L.ptr(y), L.ptr(pos), n_rows, self.k, cout, L.ptr(out), st)
            return out
        if fragments:
            L.call('u2mkd_conv_forward_pairs_f16x2' if fragments == 2 else 'u2mkd_conv_forward_pairs_x3', L.ptr(feats), n, cin, L.ptr(wt), cout, L.ptr(idx), L.ptr(self.tile_k),
                   L.ptr(self.meta), self.cap, self.k, L.ptr(y), st)
        else:
            L.call('u2mkd_conv_forward_pairs', L.ptr(feats), n, cin, L.ptr(wt), cout, L.ptr(idx), L.ptr(self.tile_k),
                   L.ptr(self.meta), self.cap, self.k, variant, L.ptr(y), st)
        if epilogue is not None:      # folded eval-mode BatchNorm (+ residual, + ReLU) at the gather-sum's store
            sc, sh, res, relu = epilogue
            L.call('u2mkd_pairs_gather_sum_ep', L.ptr(y), L.ptr(pos), n_rows, self.k, cout, L.ptr(sc), L.ptr(sh), L.ptr(res), int(relu),
                   L.ptr(out), st)
            return out
        sink = BN_STATS_SINK[0]
        if sink is not None and sink.partial is None and out.dtype == torch.float32 and _stats_in_store(n_rows, cout):
            # a train-mode BatchNorm follows (lidar.blocks.FusedSequential asked): its slab statistics in this store
            rows = int(L.load().u2mkd_pairs_gather_sum_stats_slab_rows())
            partial = torch.empty((n_rows + rows - 1) // rows * 2 * cout, dtype=torch.float32, device=out.device)
            L.call('u2mkd_pairs_gather_sum_stats', L.ptr(y), L.ptr(pos), n_rows, self.k, cout, L.ptr(out), L.ptr(partial), st)
            sink.partial, sink.slab_rows, sink.of = partial, rows, (out.data_ptr(), n_rows, cout)
            return out
        L.call('u2mkd_pairs_gather_sum', L.ptr(y), L.ptr(pos), n_rows, self.k, cout, L.ptr(out), st)
        return out


# What a network asked of its kernel maps so far: {(network tag, map key): {('schedule', inverse), 'pair_schedule', 'pairs_plan'}}.
# The tile schedule, the pair schedule and the weight gradient's compacted pairs are built at a map's FIRST use -- in the middle
# of a forward / backward pass, on the stream that pass runs on (27 such builds, ~150 small launches, per KD step on the
# student's chain).  They depend on the map only, and which of them a network needs is a property of the network (channel
# counts), not of the batch: prefetch_kmaps(tag=...) builds, next to the maps of batch k + 1, what the same network asked of the
# maps of the batches before (on the stream the geometry is prepared on; a use seen for the first time is built lazily as ever).
KMAP_USES = {}


class KernelMap:
    """Kernel map of one (tensor_stride, kernel_size, stride, dilation) key.

    ``nbr`` int32 [K, n_out]: input index feeding output row j through offset k
    (-1 = none).  ``nbr_inv`` int32 [K, n_in]: output index fed by input row i
    through offset k (strided maps; for submanifold maps it is ``nbr`` with the
    offset order reversed and is not materialised).  Indexing ``kmap[0..2]``
    gives torchsparse's (nbmaps [P,2] (in,out), nbsizes [K], (n_in, n_out)).
    """

    def __init__(self, nbr, nbr_inv, n_in, n_out, symmetric, out_coords):
        self.nbr = nbr
        self.nbr_inv = nbr_inv
        self.n_in = int(n_in)
        self.n_out = int(n_out)
        self.k = int(nbr.shape[0])
        self.symmetric = bool(symmetric)
        self.out_coords = out_coords
        self._rulebook = None
        self._pairs = None
        self._sorted = {}
        self._pair_schedule = None
        self.tag = None          # (network tag, map key): set by prefetch_kmaps(tag=...), see KMAP_USES

    def _note(self, use):
        if self.tag is not None:
            uses = KMAP_USES.get(self.tag)
            if uses is None:
                uses = KMAP_USES[self.tag] = set()
            uses.add(use)

    def prebuild(self, uses):
        """Build the derived structures in ``uses`` now, on the current stream (prefetch_kmaps: what the same network asked
        of the same map key in earlier steps)."""
        for use in sorted(uses, key=repr):
            if use == 'pair_schedule':
                self.pair_schedule()
            elif use == 'pairs_plan':
                self.pairs_plan()
            elif use[0] == 'schedule' and (not use[1] or self.nbr_inv is not None):
                self.schedule(use[1])

    def sorted_table(self, inverse=False):
        """(table with its rows permuted into schedule order, order int32 [rows])."""
        return self.schedule(inverse).tiles()

    def schedule(self, inverse=False):
        """The tile schedule of the (inverse) neighbour table (cached)."""
        hit = self._sorted.get(inverse)
        if hit is None:
            self._note(('schedule', bool(inverse)))
            hit = TileSchedule(self.nbr_inv if inverse else self.nbr)
            self._sorted[inverse] = hit
        return hit

    def pair_schedule(self):
        """The pair schedule of the map (cached; serves forward, transposed and dgrad)."""
        if self._pair_schedule is None:
            self._note('pair_schedule')
            self._pair_schedule = PairSchedule(self.nbr, self.n_in)
        return self._pair_schedule

    def pairs_plan(self):
        """(pairs int32 [cap,2] rows (in,out) grouped by offset, nbsizes int32 [K], plan int32):
        the compacted rulebook plus the device-side work split of the weight-gradient
        kernel.  Built without any host synchronisation: the pair buffer is sized by
        the upper bound K * n_out and only its first P rows are meaningful."""
        if self._pairs is None:
            self._note('pairs_plan')
            k, n_out = self.k, self.n_out
            dev = self.nbr.device
            nblocks = max((n_out + 1023) // 1024, 1)
            nbsizes = torch.zeros(k, dtype=torch.int32, device=dev)
            block_counts = torch.empty(k, nblocks, dtype=torch.int32, device=dev)
            cap = max(min(k * n_out, self.n_in * k), 1)
            pairs = torch.empty(cap, 2, dtype=torch.int32, device=dev)
            plan = torch.empty(L.load().u2mkd_wgrad_plan_ints(k), dtype=torch.int32, device=dev)
            if n_out:
                L.call('u2mkd_kmap_sizes', L.ptr(self.nbr), n_out, k, L.ptr(nbsizes), L.ptr(block_counts), L.stream())
                L.call('u2mkd_kmap_compact', L.ptr(self.nbr), n_out, k, L.ptr(nbsizes), L.ptr(block_counts),
                       L.ptr(pairs), L.stream())
            L.call('u2mkd_wgrad_plan', L.ptr(nbsizes), k, n_out, L.ptr(plan), L.stream())
            self._pairs = (pairs, nbsizes, plan)
        return self._pairs

    def rulebook(self):
        """torchsparse's (nbmaps [P,2], nbsizes [K]); trimming to P is the only host sync."""
        if self._rulebook is None:
            pairs, nbsizes, plan = self.pairs_plan()
            total = int(plan[0].item())
            self._rulebook = (pairs[:total], nbsizes)
        return self._rulebook

    def __len__(self):
        return 3

    def __getitem__(self, i):
        if i == 0:
            return self.rulebook()[0]
        if i == 1:
            return self.rulebook()[1]
        if i == 2:
            return (self.n_in, self.n_out)
        raise IndexError(i)

    def __iter__(self):
        return iter((self[0], self[1], self[2]))


def build_kmap(coords: torch.Tensor, tensor_stride, kernel_size, stride, out_coords=None) -> KernelMap:
    """Hash the input coordinates, probe every (output, offset) and fill the
    neighbour table (v1.4.0 conv3d kmap build, fused).  ``out_coords``: the strided map's output coordinates when the
    caller already holds them (DownsamplePyramid); default: spdownsample."""
    coords = _i32(coords).contiguous()
    L.require_cuda(coords)
    dev = coords.device
    n_in = coords.shape[0]
    offsets = get_kernel_offsets(kernel_size, stride=tensor_stride, device=dev)
    k = offsets.shape[0]
    table = coords_table(coords)
    strided = any(s > 1 for s in stride)
    if not strided:
        out_coords = coords
    elif out_coords is None:
        out_coords = spdownsample(coords, stride, kernel_size, tensor_stride)
    n_out = out_coords.shape[0]
    nbr = torch.empty(k, n_out, dtype=torch.int32, device=dev)
    L.call('u2mkd_kmap_build_table', L.ptr(table.buf), n_in, L.ptr(out_coords), n_out, L.ptr(offsets), k,
           L.ptr(nbr), L.stream())
    symmetric = (not strided) and (k % 2 == 1)
    nbr_inv = None
    if not symmetric:
        nbr_inv = torch.full((k, n_in), -1, dtype=torch.int32, device=dev)
        L.call('u2mkd_kmap_invert', L.ptr(nbr), n_out, k, n_in, L.ptr(nbr_inv), L.stream())
    return KernelMap(nbr, nbr_inv, n_in, n_out, symmetric, out_coords)


def prefetch_kmaps(x: SparseTensor, specs, level_coords=None, tag=None) -> None:
    """Build the kernel maps a network is about to ask for, in one go.

    ``specs`` lists (kernel_size, stride) of the convs that create maps, in forward order.
    Maps depend on coordinates only, but a strided map needs the unique down-sampled
    coordinates -- a host synchronisation (output size).  torchsparse (and this drop-in's
    ``conv3d``) build maps lazily at the first conv of each stride, i.e. the host stops
    four times in the middle of the encoder and cannot queue work ahead of the GPU there.
    Prefetching moves those stops to the start of the step, where the GPU queue is empty
    anyway; everything after is queued without waiting.  Results land in ``x.kmaps`` /
    ``x.cmaps`` exactly as the lazy path would leave them.  ``level_coords`` = {total stride: coords} from a
    DownsamplePyramid: the strided maps then take their output coordinates from it and the loop never waits.  ``tag``: the
    network these maps are for (any hashable): the schedules it asked of earlier batches' maps are built here too (KMAP_USES)."""
    coords, ts = x.coords, x.stride
    x.cmaps.setdefault(ts, coords)
    one = (1, 1, 1)
    for kernel_size, stride in specs:
        kernel_size, stride = make_ntuple(kernel_size, ndim=3), make_ntuple(stride, ndim=3)
        key = (ts, kernel_size, stride, one)
        kmap = x.kmaps.get(key)
        if kmap is None:
            nxt = None
            if level_coords is not None and stride != one:
                nxt = level_coords.get(tuple(ts[k] * stride[k] for k in range(3)))
            kmap = x.kmaps[key] = build_kmap(coords, ts, kernel_size, stride, out_coords=nxt)
            if tag is not None and _PREFETCH_PLANS and 'B' in _PREFETCH_PARTS:
                kmap.tag = (tag, key)
                kmap.prebuild(KMAP_USES.get(kmap.tag, ()))
        if stride != one:
            coords, ts = kmap.out_coords, tuple(ts[k] * stride[k] for k in range(3))
            x.cmaps.setdefault(ts, coords)


# --------------------------------------------------------------------- conv
def _conv_os(feats, weight, transpose, cout, kmap, inverse, n_rows, kflip, epilogue=None):
    """out[j] = sum_k feats[tbl[k][j]] @ B_k, tbl = the kernel map's (inverse) neighbour table, B_k taken from
    `weight` = kernel [K, cin, cout]: B_k[col][ci] = weight[kk][ci][col] (transpose = True: the forward) or
    weight[kk][col][ci] (False: the input gradient), kk = K-1-k if kflip else k.  kflip = 1 is the input
    gradient of a symmetric (submanifold) map computed on the forward table with mirrored offsets -- in the
    pair schedule that is simply the swapped-role walk."""
    out = torch.empty(n_rows, cout, dtype=feats.dtype, device=feats.device)
    if feats.dtype == torch.bfloat16 and not L.load().u2mkd_conv_tiles_supported(feats.shape[1], cout, kmap.k):
        # bf16 storage, every shape the tile kernel has no instantiation for (_conv_bf16_ok: multiples of 32)
        wt = _weight_layout(weight, transpose, True, arith=3)
        return kmap.pair_schedule().run(feats, wt, cout, bool(inverse) or bool(kflip), out)
    if feats.dtype != torch.bfloat16 and _pairs_mode(feats.shape[1], cout, n_rows):
        x3 = _PAIRS_X3 and bool(L.load().u2mkd_conv_pairs_x3_supported(feats.shape[1], cout))
        f2 = x3 and _pairs_f16x2(feats.shape[1], cout)          # the same kernel in f16x2 arithmetic (arith-4 fragments)
        wt = _weight_layout(weight, transpose, x3, arith=4 if f2 else 0)
        return kmap.pair_schedule().run(feats, wt, cout, bool(inverse) or bool(kflip), out, fragments=2 if f2 else x3, epilogue=epilogue)
    if inverse and kmap.nbr_inv is None:      # symmetric map: the inverse table is the mirrored forward table
        inverse, kflip = False, 1 - int(kflip)
    sch = kmap.schedule(inverse)
    assert sch.n == n_rows
    return sch.run(feats, weight, transpose, cout, kflip, out, epilogue=epilogue)


def conv_epilogue_supported(feats, cin, cout, k, n_rows):
    """True where conv_eval_affine can fold the affine into the convolution's store: fp32 rows on the tile-pair kernel
    (u2mkd_conv_forward_tiles_ep) or on the pair schedule (u2mkd_pairs_gather_sum_ep)."""
    if feats.dtype != torch.float32 or bf16_rows() or cin % 4 or cout % 4:
        return False
    if _pairs_mode(cin, cout, n_rows):
        return True
    return bool(L.load().u2mkd_conv_tiles_supported(cin, cout, k)) and conv_arith_is_default()


_TILES_ARITH = {}
_PAIRS_F16X2 = {}


def _pairs_f16x2(cin, cout):
    """u2mkd_conv_pairs_f16x2_supported, cached: the pair-schedule / dense kernels run this shape in f16x2 arithmetic."""
    key = (cin, cout)
    hit = _PAIRS_F16X2.get(key)
    if hit is None:
        hit = _PAIRS_F16X2[key] = bool(L.load().u2mkd_conv_pairs_f16x2_supported(cin, cout))
    return hit



def _tiles_arith(cin, cout, k):
    """u2mkd_conv_tiles_arith, cached: the arithmetic code the tile kernel runs a layer of fp32 rows in (4 = f16x2, 2 = bf16x3,
    1 = f32, 0 = not a tile-kernel layer)."""
    key = (cin, cout, k)
    hit = _TILES_ARITH.get(key)
    if hit is None:
        hit = _TILES_ARITH[key] = int(L.load().u2mkd_conv_tiles_arith(cin, cout, k))
    return hit


def conv_arith_is_default():
    return os.environ.get('U2MKD_CONV_SCHEDULE') in (None, 'tiles', 'pairs')


def conv_eval_affine(input: SparseTensor, conv, scale, shift, relu, residual=None):
    """INFERENCE: ``relu?(conv(input) * scale + shift (+ residual))`` with the affine, the add and the ReLU inside the
    convolution's last pass -- spnn.Conv3d followed by an eval-mode spnn.BatchNorm (scale = gamma / sqrt(running_var + eps),
    shift = beta - running_mean * scale; core/models/build_blocks.py:25-31,59-71) as one launch sequence instead of two: the
    frozen teacher of the KD step and every evaluation pass.  Returns None where the shape has no folded form (the caller
    then runs the two modules).  No autograd."""
    kernel_size, stride, dilation = make_ntuple(conv.kernel_size, ndim=3), make_ntuple(conv.stride, ndim=3), make_ntuple(conv.dilation, ndim=3)
    weight = conv.kernel
    if conv.bias is not None or weight.dim() != 3 or kernel_size == (1, 1, 1):
        return None
    feats = input.feats
    k, cin, cout = weight.shape
    if not conv.transposed:
        key = (input.stride, kernel_size, stride, dilation)
        kmap = input.kmaps.get(key)
        if kmap is None:
            kmap = build_kmap(input.coords, input.stride, kernel_size, stride)
            input.kmaps[key] = kmap
        inverse, n_rows = False, kmap.n_out
        out_coords, out_stride = kmap.out_coords, tuple(input.stride[i] * stride[i] for i in range(3))
    else:
        out_stride = tuple(input.stride[i] // stride[i] for i in range(3))
        kmap = input.kmaps[(out_stride, kernel_size, stride, dilation)]
        inverse, n_rows = True, kmap.n_in
        out_coords = input.cmaps[out_stride]
    if not conv_epilogue_supported(feats, cin, cout, k, n_rows) or feats.shape[1] != cin:
        return None
    res = None
    if residual is not None:
        res = (residual.F if isinstance(residual, SparseTensor) else residual).contiguous().float()
    with torch.no_grad():
        out = _conv_os(feats.contiguous(), weight.contiguous().float(), True, cout, kmap, inverse, n_rows, 0,
                       epilogue=(scale, shift, res, bool(relu)))
    output = SparseTensor(coords=out_coords, feats=out, stride=out_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output


def eval_bn_affine(bn):
    """(scale, shift) of an eval-mode BatchNorm as fp32 device vectors, cached on the module until a parameter, a running
    statistic or eps changes (versions of the five tensors)."""
    key = (bn.running_mean._version, bn.running_var._version, None if bn.weight is None else bn.weight._version,
           None if bn.bias is None else bn.bias._version, bn.eps, bn.running_mean.data_ptr())
    hit = bn.__dict__.get('_u2mkd_affine')
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        scale = inv * bn.weight.float() if bn.weight is not None else inv
        shift = (bn.bias.float() if bn.bias is not None else 0.0) - bn.running_mean.float() * scale
        scale, shift = scale.contiguous(), shift.contiguous()
    bn.__dict__['_u2mkd_affine'] = (key, scale, shift)
    return scale, shift


_WEIGHT_EPOCH = [0]


def invalidate_weight_caches(*_):
    """Forget every cached weight re-layout.  The caches are stamped with the tensor's in-place version counter and
    storage address, which writes through ``.data`` (``p.data.add_`` of an EMA or a hand-written optimizer,
    ``kernel.data.uniform_`` of a re-initialisation) do not move; so the stamp also carries this epoch, bumped after
    EVERY ``torch.optim.Optimizer.step`` (global post hook below) and by whoever writes weights behind autograd's back."""
    _WEIGHT_EPOCH[0] += 1


# Trainable weights whose MFMA-fragment images are live: (id(weight), slot) -> [weakref, slot, both, k, r, c, planes, data_ptr].
# After an optimizer step ALL of them are re-laid by ONE launch (u2mkd_weight_fragments_batch) instead of one ~5 us
# latency-bound launch per weight in front of its first convolution of the next step (~100 per KD step).
_FRAG_JOBS = {}
_FRAG_TABLE = [None, 0]            # (device job table int64 [n, 8] or None = rebuild, total units)


def _register_fragments(weight, slot, both, k, r, c, arith):
    planes = L.load().u2mkd_weight_fragments_bytes(1, 32, 32, arith) // (32 * 32 * 2)      # (f16x2: 2, + the scale trailer)
    if planes not in (1, 2, 3):    # f32 fragments (U2MKD_CONV_ARITH=f32): per-weight launches only
        return
    if planes == 2 and arith != 4:
        return                     # (f32 fragments are 2 x 2 bytes per element too)
    _FRAG_JOBS[(id(weight), slot)] = [weakref.ref(weight), slot, both, k, r, c, planes, weight.data_ptr()]
    _FRAG_TABLE[0] = None


def refresh_weight_fragments(*_):
    """Optimizer-step post hook: new epoch (see ``invalidate_weight_caches``), then every registered trainable weight's
    fragments of both orientations are rebuilt from the updated values in one launch and re-stamped, so the next
    step's convolutions find them current.  A weight that died, moved, was frozen or whose cache entry was replaced
    drops out (it re-registers through the per-weight path on its next use)."""
    _WEIGHT_EPOCH[0] += 1
    if not _FRAG_JOBS:
        return
    live = []
    for key, job in list(_FRAG_JOBS.items()):
        w = job[0]()
        hit = None if w is None else w.__dict__.get(job[1])
        if w is None or not w.requires_grad or w.data_ptr() != job[7] or hit is None or hit[1] is not job[2]:
            del _FRAG_JOBS[key]
            _FRAG_TABLE[0] = None
        else:
            live.append((w, job))
    if not live:
        return
    if _FRAG_TABLE[0] is None:
        rows, first = [], 0
        for w, (_, _, both, k, r, c, planes, ptr) in live:
            rows.append([ptr, both.data_ptr(), first, k, r, c, planes, 0])
            first += 2 * (k * r * c // 512)
        # pinned staging + asynchronous copy: a pageable copy makes the host wait for everything queued on the stream --
        # right behind the optimizer that is the whole backward, i.e. the host's ~20 ms lead over the GPU
        stage = torch.tensor(rows, dtype=torch.int64).pin_memory()
        _FRAG_TABLE[0] = stage.to(live[0][0].device, non_blocking=True)
        _FRAG_TABLE[1] = first
        _FRAG_TABLE[2:] = [stage]          # alive until the next rebuild: the copy reads it asynchronously
    L.call('u2mkd_weight_fragments_batch', L.ptr(_FRAG_TABLE[0]), len(live), _FRAG_TABLE[1], L.stream())
    epoch = _WEIGHT_EPOCH[0]
    for w, job in live:
        w.__dict__[job[1]] = ((w._version, job[7], epoch), job[2])


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post
    _reg_post(refresh_weight_fragments)
except ImportError:                                  # pragma: no cover -- torch < 2.0 has no global hook
    pass


def _conv_bf16_ok(cin, cout):
    """channel counts the bf16-storage conv kernels take (tile kernel: 32..128; pair kernel: multiples of 32)"""
    return cin >= 32 and cin % 32 == 0 and cout >= 32 and cout % 32 == 0


def _weight_layout(weight, transpose, fragments, arith=0):
    """The layout of `kernel` [K, R, C] a conv kernel reads: rows of B_k = output columns, reduction contiguous.
    transpose: B_k[col][red] = weight[k][red][col] (else weight[k][col][red], the tensor as it is).
    fragments: MFMA operand-fragment order (u2mkd_weight_fragments) instead of row-major [K, ncol, nred].
    Fragments of BOTH orientations come from one launch (latency-bound: 5 us for one or for two) and are cached
    on the tensor with its in-place version, storage address and (trainable weights) the optimizer-step epoch of
    ``invalidate_weight_caches``, so the forward's launch also serves the input gradient of the same step but never a
    later step; row-major layouts are cached for FROZEN weights only (requires_grad False: the KD teacher, inference),
    trained weights are re-laid out per call."""
    if not transpose and not fragments:
        return weight
    k, r, c = weight.shape if weight.dim() == 3 else (1,) + tuple(weight.shape)     # 2-D: nn.Linear's [out, in], one offset
    frozen = not weight.requires_grad
    stamp = (weight._version, weight.data_ptr(), _WEIGHT_EPOCH[0] if not frozen else -1)
    # the cache lives on the parameter: a reshaping view of it made per call (`conv.weight.squeeze(-1)` of a k = 1 Conv1d,
    # `.view(cout, cin)` of a 1x1 Conv2d) is a new tensor object every time and would re-lay the weight on every use
    holder, tag = weight, ''
    base = weight._base
    if base is not None and base.numel() == weight.numel() and base.data_ptr() == weight.data_ptr() and base.is_contiguous() \
            and weight.is_contiguous() and base.requires_grad == weight.requires_grad:
        holder, tag = base, ':%dx%dx%d' % (k, r, c)
    if fragments:
        # arith 0 = the library's fp32-row arithmetic (bf16x3 / f32), 3 = ONE bf16 plane (bf16 storage), 4 = f16x2 (tile kernel)
        slot = {3: '_u2mkd_wfrag3', 4: '_u2mkd_wfrag4'}.get(arith, '_u2mkd_wfrag') + tag
        hit = holder.__dict__.get(slot)
        if hit is None or hit[0] != stamp:
            nbytes = L.load().u2mkd_weight_fragments_bytes(k, r, c, arith)
            fresh = hit is None or hit[1].shape[1] != nbytes or hit[1].device != weight.device
            # (a stale image is overwritten in place: every reader is ordered on the launch stream, and the buffer's
            # address is what the batched refresh table holds)
            both = torch.empty(2, nbytes, dtype=torch.uint8, device=weight.device) if fresh else hit[1]
            L.call('u2mkd_weight_fragments', L.ptr(weight), k, r, c, 2, arith, L.ptr(both), L.stream())
            hit = (stamp, both)
            holder.__dict__[slot] = hit
            job = _FRAG_JOBS.get((id(holder), slot))
            # (leaves only: a padded / cast copy made per call dies with the call -- registering it made the refresh rebuild
            # its job table after every optimizer step)
            if not frozen and holder.is_leaf and (fresh or job is None or job[0]() is not holder or job[7] != stamp[1] or job[2] is not both):
                _register_fragments(holder, slot, both, k, r, c, arith)
        return hit[1][0 if transpose else 1]
    key = '_u2mkd_wt' + tag
    if frozen:
        hit = holder.__dict__.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
    out = torch.empty(k, c, r, dtype=torch.float32, device=weight.device)
    L.call('u2mkd_transpose_weights', L.ptr(weight), k, r, c, L.ptr(out), L.stream())
    if frozen:
        holder.__dict__[key] = (stamp, out)
    return out


def _transpose_weights(weight):
    """kernel [K, cin, cout] -> [K, cout, cin] (reduction dim contiguous for the MFMA B operand)."""
    return _weight_layout(weight, True, False)


_IDENTITY_PAIRS = {}


def _identity_pairs(n, device):
    """(pairs int32 [n,2] = (i,i), plan) of the one-offset identity map: lets the pair-list weight
    gradient kernel compute dW = dY^T X of a linear layer.  Cached per row count."""
    key = (n, device.index)
    hit = _IDENTITY_PAIRS.get(key)
    if hit is None:
        if len(_IDENTITY_PAIRS) > 64:
            _IDENTITY_PAIRS.clear()
        i = torch.arange(n, dtype=torch.int32, device=device)
        pairs = torch.stack([i, i], 1).contiguous()
        nbsizes = torch.full((1,), n, dtype=torch.int32, device=device)
        plan = torch.empty(L.load().u2mkd_wgrad_plan_ints(1), dtype=torch.int32, device=device)
        L.call('u2mkd_wgrad_plan', L.ptr(nbsizes), 1, n, L.ptr(plan), L.stream())
        hit = _IDENTITY_PAIRS[key] = (pairs, plan)
    return hit


_LINEAR_X3 = os.environ.get('U2MKD_LINEAR_X3', '1') != '0'


def _dense_x3_ok(cin, cout):
    return _PAIRS_X3 and _LINEAR_X3 and bool(L.load().u2mkd_conv_pairs_x3_supported(cin, cout))


def _dense_x3(x, weight, forward, bias=None, kernel_layout=False):
    """x @ weight.T (+ bias) (forward) or x @ weight (the input gradient) for nn.Linear's weight [out, in] on the
    bf16x3 pair kernel's dense mode; the fragment-order weights of both orientations come from one cached launch.
    bf16 rows (bf16 storage): the same kernel's one-plane form, bf16 in and out, fp32 bias and accumulation.
    ``kernel_layout``: ``weight`` is [in, out] (the one offset of a 1 x 1 x 1 spnn.Conv3d ``kernel``): x @ weight forward,
    x @ weight.T for the input gradient -- the other orientation of the same cached fragments, no transposed copy."""
    n = x.shape[0]
    if kernel_layout:          # [in, out] or the conv kernel itself, [1, in, out]
        cout = weight.shape[-1] if forward else weight.shape[-2]
    else:
        cout = weight.shape[0] if forward else weight.shape[1]
    b16 = x.dtype == torch.bfloat16
    f2 = not b16 and _pairs_f16x2(x.shape[1], cout)
    wf = _weight_layout(weight, forward if kernel_layout else not forward, True, arith=3 if b16 else (4 if f2 else 0))
    y = torch.empty(n, cout, dtype=x.dtype, device=x.device)
    L.call('u2mkd_linear_forward_bf16' if b16 else ('u2mkd_linear_forward_f16x2' if f2 else 'u2mkd_linear_forward_x3'), L.ptr(x), n,
           x.shape[1], L.ptr(wf), cout, L.ptr(bias), L.ptr(y), L.stream())
    return y


def _dense(x, w_oc_ic, bias=None):
    """rows of x [n, cin] times w^T, w = [cout, cin] (+ bias) on the MFMA pair pipeline."""
    n, cin = x.shape
    cout = w_oc_ic.shape[0]
    y = torch.empty((n + 63) // 64 * 64, cout, dtype=torch.float32, device=x.device)
    L.call('u2mkd_linear_forward', L.ptr(x), n, cin, L.ptr(w_oc_ic), cout, L.ptr(bias), 0, L.ptr(y), L.stream())
    return y[:n]


_VIEW_NODES = ('SqueezeBackward0', 'SqueezeBackward1', 'UnsqueezeBackward0', 'ViewBackward0', 'UnsafeViewBackward0', 'AliasBackward0')
_LEAF_VIEWS = os.environ.get('U2MKD_DEFER_VIEW_WGRAD', '1') != '0'      # 0: only a leaf parameter's own weight gradient is deferred


def _leaf_behind_view(w):
    """The leaf parameter ``w`` is a pure reshaping view of (same elements, same memory, contiguous both, ONE metadata-only
    autograd node between them), else None."""
    base = w._base
    fn = w.grad_fn
    if (not _LEAF_VIEWS or base is None or fn is None or not base.is_leaf or not base.requires_grad or base.numel() != w.numel()
            or base.data_ptr() != w.data_ptr() or base.dtype != w.dtype or not base.is_contiguous() or not w.is_contiguous()
            or type(fn).__name__ not in _VIEW_NODES or len(fn.next_functions) != 1):
        return None
    acc = fn.next_functions[0][0]
    if acc is None or type(acc).__name__ != 'AccumulateGrad' or getattr(acc, 'variable', None) is not base:
        return None
    return base


class LinearFunction(Function):
    """y = x @ weight.T + bias (nn.Linear semantics) on the conv pipeline: forward and input
    gradient on the pair kernel's dense mode, weight gradient on the pair-list wgrad kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, bias_grad_is_zero=False):
        L.require_cuda(x, weight)
        param = weight
        weight = weight.contiguous().float()
        # the saved weight IS the parameter (no contiguous / cast copy): its gradient goes to AccumulateGrad untouched -- or it is a
        # pure reshaping view of one (`conv.weight.squeeze(-1)` of a k = 1 Conv1d: fusion_blocks.py's Conv1d layers evaluated on
        # rows), whose gradient reaches the parameter's AccumulateGrad through a metadata-only node
        owner = param if param.is_leaf else _leaf_behind_view(param)
        ctx.weight_owner = owner if (owner is not None and weight.data_ptr() == param.data_ptr() and weight.dtype == param.dtype) else None
        ctx.weight_is_param = ctx.weight_owner is not None
        # the bias IS a leaf parameter (not a padded / cast copy): its gradient, too, is read by nobody before the backward ends
        ctx.bias_param = bias if (bias is not None and bias.is_leaf and bias.requires_grad and bias.dtype == torch.float32) else None
        ctx.overlap_ok = _deferred_overlap_ok()
        want16 = bf16_rows()
        b16 = want16 and _conv_bf16_ok(weight.shape[1], weight.shape[0])
        ctx.in_dtype = x.dtype
        x = _rows(x, b16)
        if x.dim() != 2 or x.shape[1] != weight.shape[1]:
            raise RuntimeError(f'linear: input {tuple(x.shape)} does not match weight {tuple(weight.shape)}')
        if weight.shape[1] % 4 != 0 or weight.shape[0] % 4 != 0:
            raise RuntimeError('linear: in/out features must be multiples of 4 on the HIP path')
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_grad_is_zero = bool(bias_grad_is_zero)
        if x.shape[0] == 0:
            return x.new_zeros(0, weight.shape[0])
        b = bias.contiguous().float() if bias is not None else None
        ctx.x3 = b16 or _dense_x3_ok(weight.shape[1], weight.shape[0])
        y = _dense_x3(x, weight, True, b) if ctx.x3 else _dense(x, weight, b)
        return y.to(torch.bfloat16) if (want16 and not b16) else y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        b16 = x.dtype == torch.bfloat16
        g = _rows(g, b16)
        n = x.shape[0]
        cout, cin = weight.shape
        gx = gw = gb = None
        if n == 0:
            return x.new_zeros(x.shape), torch.zeros_like(weight), (weight.new_zeros(cout) if ctx.has_bias else None), None
        side, deferred_join = None, False
        # (the weight gradient first: it goes to the side stream and runs next to gx)
        if ctx.needs_input_grad[1]:
            pairs, plan = _identity_pairs(n, g.device)
            lib = L.load()
            nbytes = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cout, cin, 1)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.device)
            gw = torch.empty_like(weight)
            side, deferred_join = _wgrad_side(ctx.weight_owner if ctx.weight_is_param else weight, ctx.weight_is_param, g.device,
                                               ctx.needs_input_grad[0], x, g, ws, gw, pairs, plan, allow=getattr(ctx, 'overlap_ok', True))
            L.call('u2mkd_conv_wgrad_pairs_bf16' if b16 else 'u2mkd_conv_wgrad_pairs', L.ptr(g), cout, L.ptr(x), cin,
                   L.ptr(pairs), L.ptr(plan), n, 1, 0, L.ptr(ws), nbytes, L.ptr(gw), side.cuda_stream if side is not None else L.stream())
        if ctx.needs_input_grad[0] and ctx.x3:
            gx = _dense_x3(g, weight, False)
        elif ctx.needs_input_grad[0]:
            w_t = torch.empty(1, cin, cout, dtype=torch.float32, device=g.device)     # [cin][cout] = W^T rows
            L.call('u2mkd_transpose_weights', L.ptr(weight), 1, cout, cin, L.ptr(w_t), L.stream())
            gx = _dense(g, w_t[0])
        if side is not None and not deferred_join:
            torch.cuda.current_stream(g.device).wait_stream(side)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # a bias that feeds a train-mode BatchNorm has the gradient sum(dY) = 0 identically (BatchNorm's input gradient
            # sums to zero over the batch: sum(x_hat) = 0); the reduction over [N, C] would compute rounding noise
            if _DEFER_BIAS_GRAD and deferred_join and side is not None and ctx.bias_param is not None and ctx.bias_param.grad is None:
                # behind the weight gradient on its side stream, joined with it when the backward ends: the column sum (or the
                # fill) leaves the backward's chain (29 reductions + 20 fills per KD step)
                gb = torch.empty(cout, dtype=torch.float32, device=g.device)
                gb.record_stream(side)
                with torch.cuda.stream(side):
                    if ctx.bias_grad_is_zero:
                        gb.zero_()
                    else:
                        torch.sum(g, 0, dtype=torch.float32, out=gb)
            else:
                gb = weight.new_zeros(cout) if ctx.bias_grad_is_zero else g.sum(0, dtype=torch.float32)
        if gx is not None and gx.dtype != ctx.in_dtype:
            gx = gx.to(ctx.in_dtype)
        return gx, gw, gb, None


class PointwiseConvFunction(Function):
    """spnn.Conv3d with a 1 x 1 x 1 kernel (the ``downsample`` branch of a ResidualBlock, core/models/build_blocks.py:66-70):
    y = x @ kernel (+ bias) with ``kernel`` [cin, cout] (v1.4.0 stores a one-offset kernel 2-D; [1, cin, cout] is taken too) AS
    THE PARAMETER STORES IT.  Rounds 2-5 handed nn.Linear's
    Function ``kernel[0].t()``: a transposed view that had to be copied, re-laid into MFMA fragments and scaled on every call (a
    temporary never sees the per-step batched fragment refresh) and whose gradient went back through a transposing node -- 16
    such layers per KD step.  Here the forward and the input gradient read the two orientations of the PARAMETER's cached
    fragments and the weight gradient is produced in the kernel's own layout (dW[0] = X^T dY: the pair-list kernel over the
    identity pairs, as LinearFunction uses it with the roles swapped)."""

    @staticmethod
    def forward(ctx, x, kernel, bias):
        L.require_cuda(x, kernel)
        ctx.weight_is_param = kernel.is_leaf
        ctx.overlap_ok = _deferred_overlap_ok()
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        want16 = bf16_rows()
        b16 = want16 and _conv_bf16_ok(cin, cout)
        ctx.in_dtype = x.dtype
        x = _rows(x, b16)
        if x.dim() != 2 or x.shape[1] != cin:
            raise RuntimeError(f'conv3d (1x1x1): input {tuple(x.shape)} does not match kernel {tuple(kernel.shape)}')
        ctx.save_for_backward(x, kernel)
        ctx.has_bias = bias is not None
        if x.shape[0] == 0:
            return x.new_zeros(0, cout)
        b = bias.contiguous().float() if bias is not None else None
        y = _dense_x3(x, kernel, True, b, kernel_layout=True)
        return y.to(torch.bfloat16) if (want16 and not b16) else y

    @staticmethod
    def backward(ctx, g):
        x, kernel = ctx.saved_tensors
        b16 = x.dtype == torch.bfloat16
        g = _rows(g, b16)
        n = x.shape[0]
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        gx = gw = gb = None
        if n == 0:
            return x.new_zeros(x.shape), torch.zeros_like(kernel), (kernel.new_zeros(cout) if ctx.has_bias else None)
        side, deferred_join = None, False
        if ctx.needs_input_grad[1]:
            pairs, plan = _identity_pairs(n, g.device)
            nbytes = L.load().u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, 1)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.device)
            gw = torch.empty_like(kernel)
            side, deferred_join = _wgrad_side(kernel, ctx.weight_is_param, g.device, ctx.needs_input_grad[0], x, g, ws, gw, pairs, plan,
                                               allow=getattr(ctx, 'overlap_ok', True))
            L.call('u2mkd_conv_wgrad_pairs_bf16' if b16 else 'u2mkd_conv_wgrad_pairs', L.ptr(x), cin, L.ptr(g), cout,
                   L.ptr(pairs), L.ptr(plan), n, 1, 0, L.ptr(ws), nbytes, L.ptr(gw), side.cuda_stream if side is not None else L.stream())
        if ctx.needs_input_grad[0]:
            gx = _dense_x3(g, kernel, False, kernel_layout=True)
        if side is not None and not deferred_join:
            torch.cuda.current_stream(g.device).wait_stream(side)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0, dtype=torch.float32)
        if gx is not None and gx.dtype != ctx.in_dtype:
            gx = gx.to(ctx.in_dtype)
        return gx, gw, gb


def linear(x, weight, bias=None, bias_feeds_batchnorm=False):
    """nn.functional.linear for [N, C] feature matrices on the HIP path (CPU tensors raise).  in / out features that
    are not multiples of 4 (the 17-class heads) are zero-padded to the next multiple of 32 / 4 -- rows of `weight` and
    entries of `bias` for the outputs, columns of `weight` and of `x` for the inputs; differentiable, the padding
    receives zero gradient and the extra output columns are dropped (as conv3d does for odd channel counts).
    ``bias_feeds_batchnorm``: the output goes straight into a BatchNorm in training mode -- the bias gradient is then
    exactly zero and is returned as such instead of being reduced from the output gradient."""
    cout, cin = weight.shape
    pin = (-cin) % 4
    pout = 0 if cout % 4 == 0 else ((-cout) % 32 if cin % 32 == 0 else (-cout) % 4)
    if pin == 0 and pout == 0:
        return LinearFunction.apply(x, weight, bias, bias_feeds_batchnorm)
    if pin:
        x = torch.nn.functional.pad(x, (0, pin))
    weight = torch.nn.functional.pad(weight, (0, pin, 0, pout))
    if bias is not None and pout:
        bias = torch.nn.functional.pad(bias, (0, pout))
    return LinearFunction.apply(x, weight, bias, bias_feeds_batchnorm)[:, :cout]


_OVERLAP_WGRAD = os.environ.get('U2MKD_OVERLAP_WGRAD', '1') != '0'
_DEFER_BIAS_GRAD = os.environ.get('U2MKD_DEFER_BIAS_GRAD', '1') != '0'      # nn.Linear's bias gradient behind its weight gradient on the side stream


def _side_stream(device):
    from ... import deferred
    return deferred.stream(device.index if device.index is not None else torch.cuda.current_device(), 'sparse_wgrad')


def _deferred_overlap_ok():
    from ... import deferred
    return deferred.overlap_ok()


def _wgrad_side(weight, is_param, device, overlap_inline, *used, allow=True):
    """(stream, deferred) for a weight-gradient launch.  The launch goes to the weight-gradient side stream; when the gradient
    is a leaf's first of this backward pass nobody reads it before the pass ends, and the launch is joined THERE
    (deferred.side_for: end-of-backward callback; ``used`` = the tensors the launch touches, kept from the allocator until
    the side stream is done with them) -- the weight gradient of a wide layer takes longer than its input gradient, and a
    linear layer's has no input-gradient launch of comparable length to hide behind; joined per function the backward's
    chain waited for every one of them.  KD step 70.4 -> 67.6 ms (three same-box pairs).  Measured the other way round
    earlier in round 4 (+1.5..4 ms): that was with 4 hardware queues and a host without lead -- see NOTES.md N8.
    Otherwise (an existing .grad is accumulated into right after the function returns): joined by the caller at its end."""
    # (allow: deferred.overlap_ok() as the FORWARD saw it -- the backward runs outside the autocast context)
    from ... import deferred as _d
    if not _OVERLAP_WGRAD or not allow or not _d.overlap_ok():
        return None, False
    # is_param: the tensor whose gradient this is IS a leaf parameter (not a contiguous / padded / cast copy made for the call:
    # the gradient of a copy runs through more backward nodes, on the main stream, before it reaches the parameter)
    from ... import deferred
    if is_param and weight.grad is None and deferred.enabled():
        if deferred.owned(id(weight)):
            # a second contribution to this leaf in one pass (a shared weight): autograd adds the two as soon as this function
            # returns, on this stream -- the first must be complete here, and this one is joined in line below
            deferred.join()
        else:
            side = deferred.side_for('sparse_wgrad', device, owner=id(weight))
            for t in used:
                t.record_stream(side)
            return side, True
    if not overlap_inline:
        return None, False
    side = _side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    return side, False


class ConvolutionFunction(Function):
    """Sparse conv forward / backward on the neighbour tables of a KernelMap."""

    @staticmethod
    def forward(ctx, input, weight, kmap, transposed=False):
        L.require_cuda(input, weight)
        param = weight
        weight = weight.contiguous().float()
        ctx.weight_is_param = param.is_leaf and weight.data_ptr() == param.data_ptr() and weight.dtype == param.dtype
        ctx.overlap_ok = _deferred_overlap_ok()
        k, cin, cout = weight.shape
        # bf16 storage (autocast to bfloat16): bf16 rows in and out, as torchsparse's custom_fwd(cast_inputs=half);
        # shapes without a bf16 kernel (the 4-channel stem) compute on fp32 rows and round the result once
        want16 = bf16_rows()
        b16 = want16 and _conv_bf16_ok(cin, cout)
        ctx.in_dtype = input.dtype
        input = _rows(input, b16)
        if input.shape[1] != cin:
            raise RuntimeError(f'conv3d: input has {input.shape[1]} channels, kernel expects {cin}')
        if cin % 4 != 0:
            raise RuntimeError(f'conv3d: in_channels={cin} must be a multiple of 4 (16-byte row gathers)')
        if not transposed:
            inverse, n_rows = False, kmap.n_out
            expect = kmap.n_in
        else:
            inverse, n_rows = True, kmap.n_in
            expect = kmap.n_out
        if input.shape[0] != expect:
            raise RuntimeError(f'conv3d: {input.shape[0]} input rows, kernel map expects {expect}')
        out = _conv_os(input, weight, True, cout, kmap, inverse, n_rows, 0)
        ctx.save_for_backward(input, weight)
        ctx.kmap = kmap
        ctx.transposed = transposed
        return out.to(torch.bfloat16) if (want16 and not b16) else out

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        kmap, transposed = ctx.kmap, ctx.transposed
        b16 = input.dtype == torch.bfloat16
        g = _rows(grad_output, b16)
        k, cin, cout = weight.shape
        grad_input = grad_weight = None
        # The two gradients are independent: the weight gradient runs on a side stream next to
        # the input gradient (each is a short grid with a long tail; together they fill the
        # chip) and is joined before this function returns, so autograd sees ordinary tensors.
        do_w = ctx.needs_input_grad[1]
        do_x = ctx.needs_input_grad[0]
        side, deferred_join = None, False
        if do_w:
            # dW[k] = sum over the offset's pairs of X[in]^T dY[out] (transposed conv: roles swapped)
            pairs, _, plan = kmap.pairs_plan()
            lib = L.load()
            nbytes = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(kmap.n_out, cin, cout, k)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.device)
            grad_weight = torch.empty_like(weight)
            side, deferred_join = _wgrad_side(weight, ctx.weight_is_param, g.device, do_x, input, g, ws, grad_weight, pairs, plan,
                                               allow=getattr(ctx, 'overlap_ok', True))
            st = side.cuda_stream if side is not None else L.stream()
            L.call('u2mkd_conv_wgrad_pairs_bf16' if b16 else 'u2mkd_conv_wgrad_pairs', L.ptr(input), cin, L.ptr(g), cout,
                   L.ptr(pairs), L.ptr(plan), kmap.n_out, k, 1 if transposed else 0, L.ptr(ws), nbytes, L.ptr(grad_weight), st)
        if do_x:
            # dX[i] = sum_k dY[out_k(i)] @ W[k]^T : same kernel on the swapped-role table,
            # B_k = W[k] read as [cin][cout] (reduction over cout contiguous).
            if not transposed:
                inverse, kflip = (False, 1) if kmap.symmetric else (True, 0)
            else:
                inverse, kflip = False, 0
            if cout % 4 != 0:
                raise RuntimeError(f'conv3d backward: out_channels={cout} must be a multiple of 4')
            grad_input = _conv_os(g, weight, False, cin, kmap, inverse, input.shape[0], kflip)
        if side is not None and not deferred_join:
            torch.cuda.current_stream(g.device).wait_stream(side)
        if grad_input is not None and grad_input.dtype != ctx.in_dtype:
            grad_input = grad_input.to(ctx.in_dtype)
        return grad_input, grad_weight, None, None


def _conv_any_channels(feats, weight, kmap, transposed):
    """ConvolutionFunction for any channel count, as torchsparse v1.4.0 accepts (e.g. in_channel = 5 with a time
    channel, or 3): the kernels move 16-byte row segments, so channel counts that are not multiples of 4 are
    zero-padded to the next multiple (rows of `kernel` for cin, columns for cout; differentiable, the padded
    entries receive zero gradient and the extra output columns are dropped)."""
    cin, cout = weight.shape[1], weight.shape[2]
    pin, pout = (-cin) % 4, (-cout) % 4
    if feats.shape[1] != cin:
        raise RuntimeError(f'conv3d: input has {feats.shape[1]} channels, kernel expects {cin}')
    if pin == 0 and pout == 0:
        return ConvolutionFunction.apply(feats, weight, kmap, transposed)
    feats = torch.nn.functional.pad(feats, (0, pin))
    weight = torch.nn.functional.pad(weight, (0, pout, 0, pin))
    return ConvolutionFunction.apply(feats, weight, kmap, transposed)[:, :cout]


def conv3d(input: SparseTensor, weight: torch.Tensor, kernel_size, bias=None, stride=1, dilation=1,
           transposed: bool = False) -> SparseTensor:
    feats, coords = input.feats, input.coords
    kernel_size = make_ntuple(kernel_size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)

    if kernel_size == (1, 1, 1) and stride == (1, 1, 1) and dilation == (1, 1, 1):
        w = weight[0] if weight.dim() == 3 else weight          # kernel [1, cin, cout] or [cin, cout]
        if (weight.dim() == 2 or weight.shape[0] == 1) and weight.is_contiguous() and weight.dtype == torch.float32 \
                and _dense_x3_ok(w.shape[0], w.shape[1]) and feats.is_cuda and feats.dim() == 2 and feats.shape[1] == w.shape[0]:
            # the kernel in its own [1, cin, cout] layout: cached fragments, no transposed copy (PointwiseConvFunction)
            feats = PointwiseConvFunction.apply(feats, weight, bias)
        elif w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0:
            feats = linear(feats, w.t(), bias)
        else:
            feats = feats.matmul(w)
            if bias is not None:
                feats = feats + bias
        output = SparseTensor(coords=coords, feats=feats, stride=input.stride)
    elif not transposed:
        key = (input.stride, kernel_size, stride, dilation)
        kmap = input.kmaps.get(key)
        if kmap is None:
            # v1.4.0 builds the offsets from the tensor stride only (dilation is not applied)
            kmap = build_kmap(coords, input.stride, kernel_size, stride)
            input.kmaps[key] = kmap
        feats = _conv_any_channels(feats, weight, kmap, transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=kmap.out_coords, feats=feats,
                              stride=tuple(input.stride[k] * stride[k] for k in range(3)))
    else:
        tensor_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        kmap = input.kmaps[(tensor_stride, kernel_size, stride, dilation)]
        feats = _conv_any_channels(feats, weight, kmap, transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=input.cmaps[tensor_stride], feats=feats, stride=tensor_stride)

    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output


# --------------------------------------------------------------- batch norm
class BatchNormFunction(Function):
    """BatchNorm over the rows of [N, C] (+ optional fused ReLU) on the HIP kernels of
    csrc/bn.hip; statistics identical to nn.BatchNorm1d."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, counter=None, res=None, stats=None):
        L.require_cuda(x)
        # bf16 storage: rows stay bf16 through the BatchNorm (as nn.BatchNorm1d passes half through under the
        # reference's amp), statistics / parameters / gradient sums fp32
        b16 = bf16_rows() or (x.dtype == torch.bfloat16 and _BF16_ROWS)
        sfx = '_bf16' if b16 else ''
        ctx.in_dtype, ctx.res_dtype = x.dtype, (res.dtype if res is not None else None)
        x = _rows(x, b16)
        n, c = x.shape
        dev = x.device
        if res is not None:      # y = relu(bn(x) + res): the tail of a ResidualBlock in the same pass
            assert relu and res.shape == x.shape, (relu, res.shape, x.shape)
            res = _rows(res, b16)
        y = torch.empty_like(x)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        if training and stats is not None and not b16:
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            L.call('u2mkd_bn_train_forward_from_partial', L.ptr(x), L.ptr(res), n, c, L.ptr(gamma), L.ptr(beta), float(eps),
                   float(momentum), L.ptr(running_mean), L.ptr(running_var), L.ptr(counter), int(relu), L.ptr(stats.partial),
                   int(stats.slab_rows), L.ptr(mean), L.ptr(invstd), L.ptr(y), L.stream())
        elif training:
            slabs = L.load().u2mkd_bn_num_slabs(n)
            partial = torch.empty(max(slabs, 1) * 2 * c, dtype=torch.float32, device=dev)
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            L.call('u2mkd_bn_train_forward_res' + sfx, L.ptr(x), L.ptr(res), n, c, L.ptr(gamma), L.ptr(beta), float(eps),
                   float(momentum), L.ptr(running_mean), L.ptr(running_var), L.ptr(counter), int(relu), L.ptr(partial),
                   L.ptr(mean), L.ptr(invstd), L.ptr(y), L.stream())
        else:
            mean = running_mean
            L.call('u2mkd_bn_eval_forward_res' + sfx, L.ptr(x), L.ptr(res), n, c, L.ptr(gamma), L.ptr(beta), float(eps),
                   L.ptr(running_mean), L.ptr(running_var), int(relu), L.ptr(invstd), L.ptr(y), L.stream())
        ctx.save_for_backward(x, gamma, beta, mean, invstd, res)
        ctx.relu, ctx.training = bool(relu), bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, res = ctx.saved_tensors
        b16 = x.dtype == torch.bfloat16
        dy = _rows(dy, b16)
        n, c = x.shape
        dev = x.device
        slabs = L.load().u2mkd_bn_num_slabs(n)
        partial = torch.empty(max(slabs, 1) * 2 * c, dtype=torch.float32, device=dev)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if res is not None else None
        L.call('u2mkd_bn_backward_res_bf16' if b16 else 'u2mkd_bn_backward_res', L.ptr(dy), L.ptr(x), L.ptr(res), n, c,
               L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), int(ctx.relu), int(ctx.training), L.ptr(partial),
               L.ptr(dgamma), L.ptr(dbeta), L.ptr(dx), L.ptr(dres), L.stream())
        if dx.dtype != ctx.in_dtype:
            dx = dx.to(ctx.in_dtype)
        if dres is not None and dres.dtype != ctx.res_dtype:
            dres = dres.to(ctx.res_dtype)
        return (dx, dgamma if gamma is not None else None, dbeta if beta is not None else None,
                None, None, None, None, None, None, None, dres, None)


# SyncBatchNorm exchanges of the current process since the last reset: [calls, bytes sent per rank] per kind.  Counted where a
# synchronising BatchNorm WOULD exchange (also at world size 1 under U2MKD_FORCE_SYNC_BN, where nothing is sent), so
# bench.py can state the collectives per step of the N > 1 path before anybody has N > 1 GPUs (distributed.collective_counts).
COLLECTIVES = {'all_gather': [0, 0], 'all_reduce': [0, 0]}


def note_collective(kind, t):
    c = COLLECTIVES[kind]
    c[0] += 1
    c[1] += t.numel() * t.element_size()


def _gather_rows(out, row, group):
    """out[r] = rank r's `row` (device tensors).  RCCL: one all_gather on the device.  A gloo group cannot move
    device tensors in an all_gather: the [2C+1] floats travel through host memory (the TRANSPORT only -- this is how
    two ranks on ONE GPU can run the device path in tests/test_gpu_sync_bn_two_ranks.py, RCCL refuses two ranks on a
    device)."""
    import torch.distributed as dist
    if dist.get_backend(group) == 'gloo':
        rows = [torch.empty(row.shape, dtype=row.dtype) for _ in range(out.shape[0])]
        dist.all_gather(rows, row.cpu(), group=group)
        out.copy_(torch.stack(rows), non_blocking=False)
    else:
        dist.all_gather_into_tensor(out, row, group=group)


def _sum_over_ranks(t, group):
    import torch.distributed as dist
    if dist.get_backend(group) == 'gloo' and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


class SyncBatchNormFunction(Function):
    """SyncBatchNorm (+ fused ReLU) over the rows of [N, C] across the ranks of a process group:
    local slab statistics -> ONE all_gather of [2C+1] floats -> Chan merge in rank order ->
    normalise; backward: local sums -> ONE all_reduce of [2C] floats -> apply with the global count.
    Same statistics as torch.nn.SyncBatchNorm (utils.py:138-220 converts every BatchNorm to it,
    train_spformer.py:79), three kernels + one small collective per pass instead of torch's
    stats / gather / elemt / ReLU kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, group, world, res=None, counter=None):
        import torch.distributed as dist
        L.require_cuda(x)
        b16 = bf16_rows() or (x.dtype == torch.bfloat16 and _BF16_ROWS)
        sfx = '_bf16' if b16 else ''
        ctx.in_dtype = x.dtype
        x = _rows(x, b16)
        if res is not None:          # relu(bn(x) + res): the tail of a ResidualBlock inside the apply pass
            ctx.res_dtype = res.dtype
            res = _rows(res, b16)
        n, c = x.shape
        dev = x.device
        st = L.stream()
        slabs = L.load().u2mkd_bn_num_slabs(n)
        partial = torch.empty(max(slabs, 1) * 2 * c, dtype=torch.float32, device=dev)
        stats = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
        L.call('u2mkd_bn_local_stats' + sfx, L.ptr(x), n, c, L.ptr(partial), L.ptr(stats), st)
        note_collective('all_gather', stats)
        if world > 1:
            gathered = torch.empty(world, 2 * c + 1, dtype=torch.float32, device=dev)
            _gather_rows(gathered, stats, group)
        else:
            gathered = stats.view(1, -1)          # (a one-rank group: the row is its own gathering)
        # mean | invstd | total in one allocation; the step counter is bumped inside the merge launch
        mit = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
        mean, invstd, total = mit[:c], mit[c:2 * c], mit[2 * c:]
        L.call('u2mkd_bn_merge_stats_counted', L.ptr(gathered), world, c, float(eps), float(momentum), L.ptr(running_mean),
               L.ptr(running_var), L.ptr(mean), L.ptr(invstd), L.ptr(total), L.ptr(counter), L.stream())
        y = torch.empty_like(x)
        if res is None:
            L.call('u2mkd_bn_apply' + sfx, L.ptr(x), n, c, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), int(relu),
                   L.ptr(y), L.stream())
        else:
            L.call('u2mkd_bn_apply_res' + sfx, L.ptr(x), L.ptr(res), n, c, L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                   L.ptr(beta), int(relu), L.ptr(y), L.stream())
        ctx.save_for_backward(x, gamma, beta, mean, invstd, total, res)
        ctx.relu, ctx.group, ctx.world = bool(relu), group, world
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, gamma, beta, mean, invstd, total, res = ctx.saved_tensors
        b16 = x.dtype == torch.bfloat16
        sfx = '_bf16' if b16 else ''
        dy = _rows(dy, b16)
        n, c = x.shape
        dev = x.device
        slabs = L.load().u2mkd_bn_num_slabs(n)
        partial = torch.empty(max(slabs, 1) * 2 * c, dtype=torch.float32, device=dev)
        # the local sums twice: `sums` goes into the all_reduce in place, `local` stays this rank's (the parameter gradients:
        # DDP averages them)
        both = torch.empty(2, 2 * c, dtype=torch.float32, device=dev)
        sums, local = both[0], both[1]
        L.call('u2mkd_bn_backward_local_keep', L.ptr(dy), L.ptr(x), L.ptr(res), int(b16), n, c, L.ptr(mean), L.ptr(invstd),
               L.ptr(gamma), L.ptr(beta), int(ctx.relu), L.ptr(partial), L.ptr(sums), L.ptr(local), L.stream())
        note_collective('all_reduce', sums)
        if ctx.world > 1:
            _sum_over_ranks(sums, ctx.group)
        dx = torch.empty_like(x)
        dres = None
        if res is None:
            L.call('u2mkd_bn_backward_apply' + sfx, L.ptr(dy), L.ptr(x), n, c, L.ptr(total), L.ptr(mean), L.ptr(invstd),
                   L.ptr(gamma), L.ptr(beta), int(ctx.relu), L.ptr(sums), L.ptr(dx), L.stream())
        else:
            dres = torch.empty_like(x)
            L.call('u2mkd_bn_backward_apply_res' + sfx, L.ptr(dy), L.ptr(x), L.ptr(res), n, c, L.ptr(total), L.ptr(mean),
                   L.ptr(invstd), L.ptr(gamma), L.ptr(beta), int(ctx.relu), L.ptr(sums), L.ptr(dx), L.ptr(dres), L.stream())
            if dres.dtype != ctx.res_dtype:
                dres = dres.to(ctx.res_dtype)
        if dx.dtype != ctx.in_dtype:
            dx = dx.to(ctx.in_dtype)
        return (dx, local[c:] if gamma is not None else None, local[:c] if beta is not None else None,
                None, None, None, None, None, None, None, dres, None)


def _sync_group(bn):
    """(process group, world size) if `bn` is a SyncBatchNorm that has to synchronise, else None."""
    if not (isinstance(bn, torch.nn.SyncBatchNorm) and bn.training):
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    if world > 1 or os.environ.get('U2MKD_FORCE_SYNC_BN') == '1':     # (the env knob: single-rank tests)
        return group, world
    return None


def batch_norm(x: torch.Tensor, bn: torch.nn.modules.batchnorm._BatchNorm, relu: bool = False,
               residual: torch.Tensor = None, stats: 'BnStats' = None) -> torch.Tensor:
    """nn.BatchNorm1d semantics (training or eval, running statistics, momentum=None =
    cumulative average) on a [N, C] tensor, optionally fused with ReLU; ``residual`` [N, C] (with relu):
    relu(bn(x) + residual), the tail of a ResidualBlock (build_blocks.py:80-83), in the same pass."""
    if x.dim() != 2:
        raise RuntimeError(f'batch_norm expects [N, C] features, got {tuple(x.shape)}')
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    factor = 0.0 if bn.momentum is None else bn.momentum
    sync = _sync_group(bn) if training else None
    counter = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        # the step counter is bumped inside the statistics / merge kernel (no launch of its own) unless its value is
        # needed on the host (momentum=None: cumulative average)
        if bn.momentum is None or x.shape[0] < 2 or not bn.num_batches_tracked.is_cuda:
            bn.num_batches_tracked.add_(1)
            if bn.momentum is None:
                factor = 1.0 / float(bn.num_batches_tracked)
        else:
            counter = bn.num_batches_tracked
    rm = bn.running_mean if (not training or bn.track_running_stats) else None
    rv = bn.running_var if (not training or bn.track_running_stats) else None
    if residual is not None and not relu:
        raise ValueError('batch_norm: a residual input is fused together with the ReLU only')
    if sync is not None:
        return SyncBatchNormFunction.apply(x, bn.weight, bn.bias, rm, rv, factor, bn.eps, relu, sync[0], sync[1], residual, counter)
    if training and x.shape[0] < 2:
        raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(x.shape)}')
    if stats is not None and training and stats.describes(x) and not bf16_rows() and (residual is None or residual.dtype == torch.float32):
        # the slab statistics came with x (the producing convolution's store): merge + apply, no statistics pass
        return BatchNormFunction.apply(x, bn.weight, bn.bias, rm, rv, training, factor, bn.eps, relu, counter, residual, stats)
    h = _HOST if _HOST is not False else host_ops()
    if h is not None and x.dtype == torch.float32 and x.is_cuda and x.shape[0] > 0 and not bf16_rows() \
            and (residual is None or residual.dtype == torch.float32):
        # the same pass with its host side in C++ (csrc_host/host_ops.cpp: BatchNormRows)
        return h.batch_norm_rows(x, bn.weight, bn.bias, rm, rv, training, factor, bn.eps, relu, counter, residual)
    return BatchNormFunction.apply(x, bn.weight, bn.bias, rm, rv, training, factor, bn.eps, relu, counter, residual)
