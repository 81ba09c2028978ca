"""CPU restatement of the torchsparse v1.4.0 operators used by U2MKD.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  torchsparse v1.4.0 is an
un-vendored dependency of the reference (README.md:44-48); its source is not
in /root/reference, so every function below restates the *published* v1.4.0
algorithm and cites the reference call site that depends on it.  Parity at
this boundary is **unpinned** by the reference; the pins are the dense
identities in tests/test_oracle_torchsparse.py.

Integer / index work is numpy (bit-exact semantics); floating point work is
torch CPU fp32/fp64 in the same operation order as the v1.4.0 CPU backend
(per kernel offset: gather -> mm -> scatter-add).
"""
from __future__ import annotations

import numpy as np
import torch

__all__ = [
    'make_ntuple', 'get_kernel_offsets', 'sphash', 'sphashquery', 'spcount',
    'spdownsample', 'build_kmap', 'conv_forward', 'conv_backward',
    'voxelize_forward', 'voxelize_backward', 'devoxelize_forward',
    'devoxelize_backward', 'calc_ti_weights', 'sparse_quantize',
]

_FNV_OFFSET = np.uint64(14695981039346656037)
_FNV_PRIME = np.uint64(1099511628211)
_MASK60 = np.uint64(0x0FFFFFFFFFFFFFFF)


def make_ntuple(x, ndim=3):
    """torchsparse.utils.make_ntuple (used at core/datasets/...:17-18)."""
    if isinstance(x, (int, np.integer)):
        return tuple(int(x) for _ in range(ndim))
    x = tuple(int(v) for v in x)
    assert len(x) == ndim
    return x


def get_kernel_offsets(size, stride=1, dilation=1):
    """torchsparse.nn.utils.get_kernel_offsets (called core/models/utils.py:84).

    Odd kernel volume: x fastest (k = (z+1)*9 + (y+1)*3 + (x+1));
    even: z fastest (k = x*4 + y*2 + z).  int32 [K, 3].
    """
    size = make_ntuple(size)
    stride = make_ntuple(stride)
    dilation = make_ntuple(dilation)
    axes = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k]
            for k in range(3)]
    if int(np.prod(size)) % 2 == 1:
        offs = [[x, y, z] for z in axes[2] for y in axes[1] for x in axes[0]]
    else:
        offs = [[x, y, z] for x in axes[0] for y in axes[1] for z in axes[2]]
    return np.asarray(offs, dtype=np.int32).reshape(-1, 3)


def _fnv(c4: np.ndarray) -> np.ndarray:
    """FNV-1a-64 over 4 int32 words, folded to 60 bits (v1.4.0 hash kernel)."""
    h = np.full(c4.shape[:-1], _FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over='ignore'):
        for j in range(4):
            h = h ^ c4[..., j].astype(np.uint32).astype(np.uint64)
            h = h * _FNV_PRIME
    h = (h >> np.uint64(60)) ^ (h & _MASK60)
    return h.astype(np.int64)


def sphash(coords: np.ndarray, offsets: np.ndarray | None = None) -> np.ndarray:
    """F.sphash (core/models/utils.py:19,43,49,86,92,133-134).

    coords int32 [N,4] = (x,y,z,b); offsets int32 [K,3] -> int64 [N] / [K,N].
    """
    coords = np.ascontiguousarray(coords)
    assert coords.dtype == np.int32 and coords.ndim == 2 and coords.shape[1] == 4
    if offsets is None:
        return _fnv(coords)
    offsets = np.ascontiguousarray(offsets)
    assert offsets.dtype == np.int32 and offsets.ndim == 2 and offsets.shape[1] == 3
    c = np.broadcast_to(coords[None], (offsets.shape[0],) + coords.shape).copy()
    with np.errstate(over='ignore'):
        c[:, :, :3] += offsets[:, None, :]
    return _fnv(c)


def sphashquery(queries: np.ndarray, references: np.ndarray) -> np.ndarray:
    """F.sphashquery (core/models/utils.py:21,50,93,135).

    Index of each query hash in ``references`` (first occurrence wins, as
    dense_hash_map::insert does not overwrite), -1 on miss; query shape kept.
    """
    q = np.asarray(queries, dtype=np.int64)
    r = np.asarray(references, dtype=np.int64).reshape(-1)
    out = np.full(q.size, -1, dtype=np.int64)
    if r.size == 0 or q.size == 0:
        return out.reshape(q.shape)
    order = np.argsort(r, kind='stable')
    rs = r[order]
    pos = np.searchsorted(rs, q.reshape(-1), side='left')
    pos_c = np.minimum(pos, r.size - 1)
    hit = rs[pos_c] == q.reshape(-1)
    out[hit] = order[pos_c[hit]]
    return out.reshape(q.shape)


def spcount(idx: np.ndarray, num: int) -> np.ndarray:
    """F.spcount (core/models/utils.py:22,51): histogram of idx >= 0, int32."""
    idx = np.asarray(idx).reshape(-1)
    v = idx[(idx >= 0) & (idx < num)]
    return np.bincount(v, minlength=num).astype(np.int32)


def spdownsample(coords: np.ndarray, stride=2, kernel_size=2, tensor_stride=1) -> np.ndarray:
    """F.spdownsample for stride[k] in {1, kernel_size[k]} (k2s2 in
    core/models/build_blocks.py:25-29).  Output sorted by (b,x,y,z)."""
    stride = make_ntuple(stride)
    kernel_size = make_ntuple(kernel_size)
    tensor_stride = make_ntuple(tensor_stride)
    assert all(stride[k] in (1, kernel_size[k]) for k in range(3)), \
        'only the stride in {1, kernel_size} branch is on the U2MKD hot path'
    ss = np.asarray([stride[k] * tensor_stride[k] for k in range(3)], dtype=np.int32)
    c = coords.copy()
    c[:, :3] = np.floor_divide(c[:, :3], ss) * ss
    c = c[:, [3, 0, 1, 2]]
    c = np.unique(c, axis=0)
    return np.ascontiguousarray(c[:, [1, 2, 3, 0]])


def build_kmap(in_coords: np.ndarray, in_stride, kernel_size, stride, out_coords=None):
    """Kernel map of F.conv3d (v1.4.0 nn/functional/conv.py, non-transposed
    branch).  Returns (nbmaps int32 [P,2] rows (in_idx,out_idx) grouped by
    kernel offset with ascending out index, nbsizes int32 [K], out_coords,
    results int64 [K,N_out])."""
    kernel_size = make_ntuple(kernel_size)
    stride = make_ntuple(stride)
    in_stride = make_ntuple(in_stride)
    offsets = get_kernel_offsets(kernel_size, stride=in_stride)
    references = sphash(in_coords)
    if out_coords is None:
        if any(s > 1 for s in stride):
            out_coords = spdownsample(in_coords, stride, kernel_size, in_stride)
        else:
            out_coords = in_coords
    queries = sphash(out_coords, offsets)
    results = sphashquery(queries, references)            # [K, N_out]
    nbsizes = (results != -1).sum(1).astype(np.int32)
    kk, jj = np.nonzero(results != -1)                     # row-major: by k, then j
    nbmaps = np.stack([results[kk, jj], jj], 1).astype(np.int32)
    return nbmaps, nbsizes, out_coords, results


def _to_t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def conv_forward(feats, weight, nbmaps, nbsizes, sizes, transposed=False):
    """ConvolutionFunction.forward (v1.4.0): out = zeros; per offset k:
    out[out_idx] += feats[in_idx] @ W[k].  ``transposed`` swaps the roles of
    the two nbmaps columns and uses sizes[0] rows."""
    feats = _to_t(feats)
    weight = _to_t(weight)
    nb = _to_t(nbmaps).long()
    n_out = sizes[0] if transposed else sizes[1]
    out = torch.zeros(n_out, weight.shape[-1], dtype=feats.dtype)
    cur = 0
    for k, n in enumerate(np.asarray(nbsizes).tolist()):
        if n == 0:
            continue
        m = nb[cur:cur + n]
        cur += n
        i_in, i_out = (m[:, 1], m[:, 0]) if transposed else (m[:, 0], m[:, 1])
        out.index_add_(0, i_out, feats[i_in] @ weight[k])
    return out


def conv_backward(feats, weight, grad_out, nbmaps, nbsizes, transposed=False):
    """ConvolutionFunction.backward (v1.4.0): per offset k
    dIn[in_idx] += dOut[out_idx] @ W[k]^T ; dW[k] = In[in_idx]^T @ dOut[out_idx]."""
    feats = _to_t(feats)
    weight = _to_t(weight)
    grad_out = _to_t(grad_out)
    nb = _to_t(nbmaps).long()
    g_in = torch.zeros_like(feats)
    g_w = torch.zeros_like(weight)
    cur = 0
    for k, n in enumerate(np.asarray(nbsizes).tolist()):
        if n == 0:
            continue
        m = nb[cur:cur + n]
        cur += n
        i_in, i_out = (m[:, 1], m[:, 0]) if transposed else (m[:, 0], m[:, 1])
        x = feats[i_in]
        g = grad_out[i_out]
        g_in.index_add_(0, i_in, g @ weight[k].t())
        g_w[k] = x.t() @ g
    return g_in, g_w


def voxelize_forward(feats, idx, counts):
    """F.spvoxelize fwd (core/models/utils.py:24,26,58):
    out[idx[i]] += feats[i] / counts[idx[i]]."""
    feats = _to_t(feats)
    idx = _to_t(idx).long().reshape(-1)
    counts = _to_t(counts).reshape(-1)
    nv = counts.shape[0]
    out = torch.zeros(nv, feats.shape[1], dtype=feats.dtype)
    ok = (idx >= 0) & (idx < nv)
    ii = idx[ok]
    ok2 = counts[ii] > 0
    sel = ok.nonzero().squeeze(1)[ok2]
    ii = ii[ok2]
    out.index_add_(0, ii, feats[sel] / counts[ii].to(feats.dtype).unsqueeze(1))
    return out


def voxelize_backward(grad_out, idx, counts, n_points):
    """F.spvoxelize bwd: g_feats[i] = g_out[idx[i]] / counts[idx[i]]."""
    grad_out = _to_t(grad_out)
    idx = _to_t(idx).long().reshape(-1)
    counts = _to_t(counts).reshape(-1)
    nv = counts.shape[0]
    g = torch.zeros(n_points, grad_out.shape[1], dtype=grad_out.dtype)
    ok = (idx >= 0) & (idx < nv)
    ii = idx[ok]
    ok2 = counts[ii] > 0
    sel = ok.nonzero().squeeze(1)[ok2]
    ii = ii[ok2]
    g[sel] = grad_out[ii] / counts[ii].to(grad_out.dtype).unsqueeze(1)
    return g


def devoxelize_forward(feats, idx, weights):
    """F.spdevoxelize fwd (core/models/utils.py:99,111):
    out[i] = sum_k [idx[i,k] >= 0] w[i,k] * feats[idx[i,k]], k ascending."""
    feats = _to_t(feats)
    idx = _to_t(idx).long()
    weights = _to_t(weights)
    out = torch.zeros(idx.shape[0], feats.shape[1], dtype=feats.dtype)
    for k in range(idx.shape[1]):
        ok = idx[:, k] >= 0
        cur = torch.zeros_like(out)
        cur[ok] = feats[idx[ok, k]]
        out += weights[:, k:k + 1] * cur
    return out


def devoxelize_backward(grad_out, idx, weights, n_vox):
    """F.spdevoxelize bwd: g_feats[idx[i,k]] += w[i,k] * g_out[i]."""
    grad_out = _to_t(grad_out)
    idx = _to_t(idx).long()
    weights = _to_t(weights)
    g = torch.zeros(n_vox, grad_out.shape[1], dtype=grad_out.dtype)
    for k in range(idx.shape[1]):
        ok = idx[:, k] >= 0
        g.index_add_(0, idx[ok, k], weights[ok, k:k + 1] * grad_out[ok])
    return g


def calc_ti_weights(coords, idx_query, scale=1):
    """F.calc_ti_weights (core/models/utils.py:94): 8 trilinear weights in
    get_kernel_offsets(2) corner order (x-major, z fastest), zeroed where
    idx == -1, renormalised.  coords f32 [N,>=3]; idx_query [8,N] -> [8,N]."""
    p = _to_t(coords)
    idx_query = _to_t(idx_query)
    if scale != 1:
        pf = torch.floor(p / scale) * scale
    else:
        pf = torch.floor(p)
    pc = pf + scale
    x, y, z = (p[:, i].view(-1, 1) for i in range(3))
    xf, yf, zf = (pf[:, i].view(-1, 1).float() for i in range(3))
    xc, yc, zc = (pc[:, i].view(-1, 1).float() for i in range(3))
    w0 = (xc - x) * (yc - y) * (zc - z)
    w1 = (xc - x) * (yc - y) * (z - zf)
    w2 = (xc - x) * (y - yf) * (zc - z)
    w3 = (xc - x) * (y - yf) * (z - zf)
    w4 = (x - xf) * (yc - y) * (zc - z)
    w5 = (x - xf) * (yc - y) * (z - zf)
    w6 = (x - xf) * (y - yf) * (zc - z)
    w7 = (x - xf) * (y - yf) * (z - zf)
    w = torch.cat([w0, w1, w2, w3, w4, w5, w6, w7], dim=1).transpose(1, 0).contiguous()
    if scale != 1:
        w = w / scale ** 3
    w[idx_query == -1] = 0
    w = w / (torch.sum(w, dim=0) + 1e-8)
    return w


def sparse_quantize(coords, voxel_size=1, return_index=False, return_inverse=False):
    """torchsparse.utils.quantize.sparse_quantize
    (core/datasets/lc_semantic_nusc_tsd_full.py:214,421)."""
    if isinstance(voxel_size, (float, int)):
        voxel_size = (voxel_size,) * 3
    vs = np.asarray(voxel_size, dtype=np.float64)
    c = np.floor(coords / vs).astype(np.int32)
    x = c - c.min(0)
    xmax = x.max(0).astype(np.uint64) + np.uint64(1)
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for k in range(x.shape[1] - 1):
        h += x[:, k].astype(np.uint64)
        h *= xmax[k + 1]
    h += x[:, -1].astype(np.uint64)
    _, indices, inverse = np.unique(h, return_index=True, return_inverse=True)
    c = c[indices]
    outs = [c]
    if return_index:
        outs.append(indices)
    if return_inverse:
        outs.append(inverse)
    return outs[0] if len(outs) == 1 else outs
