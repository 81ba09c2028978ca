"""CPU restatement of the sptr (SparseTransformer) window attention used by
SphereFormer (TEST INFRASTRUCTURE; rows a10-a11 of SURVEY.md §8a, Appendix B).

Follows, with citations relative to /root/reference:
  * window clustering  third_party/SparseTransformer/sptr/utils.py:19-78
    (torch_geometric 1.7.2 ``voxel_grid`` -> torch_cluster 1.6.3 ``grid_cluster``,
    un-vendored; restated from their published algorithm: integer key
    sum_d trunc((p_d - start_d) / size_d) * stride_d over (x, y, z, batch), start = min
    over the whole batch);
  * pair lists         src/sptr/precompute/precompute_cuda_kernel.cu:4-22;
  * scores             src/sptr/rpe/relative_pos_encoding_cuda_kernel.cu:116-139;
  * CSR softmax        sptr/utils.py:80-95 (torch_scatter segment_csr max/sum);
  * output             src/sptr/rpe/relative_pos_encoding_cuda_kernel.cu:151-174;
  * glue               sptr/modules.py:11-66.
Pinned on the reference's known-answer fixture test/test_precompute_all.py:9-19 and on
a brute-force dense per-window attention (tests/test_oracle_sptr.py).  Gradients come
from torch autograd over this forward (the CUDA backward kernels implement the same
analytical derivatives).
"""
from __future__ import annotations

import numpy as np
import torch

__all__ = ['grid_cluster', 'precompute_all', 'get_indices_params', 'relative_position_index',
           'exponential_split', 'cart2sphere', 'sparse_self_attention', 'dense_window_attention']


def grid_cluster(pos: torch.Tensor, batch: torch.Tensor, size, start=None) -> torch.Tensor:
    """voxel_grid(pos, batch, size, start): int64 cluster key per token."""
    pos = pos.float()
    size = torch.as_tensor(size, dtype=pos.dtype).reshape(-1)
    p = torch.cat([pos, batch.reshape(-1, 1).to(pos.dtype)], 1)
    sz = torch.cat([size, torch.ones(1, dtype=pos.dtype)])
    if start is None:
        st = p.min(0)[0]
    else:
        st = torch.cat([torch.as_tensor(start, dtype=pos.dtype).reshape(-1), torch.zeros(1, dtype=pos.dtype)])
    en = p.max(0)[0]
    c = torch.zeros(p.shape[0], dtype=torch.int64)
    k = 1
    for d in range(p.shape[1]):
        c += ((p[:, d] - st[d]) / sz[d]).to(torch.int64) * k
        k *= int(((en[d] - st[d]) / sz[d]).to(torch.int64)) + 1
    return c


def precompute_all(counts: np.ndarray):
    """precompute_all_cuda_kernel: pair m = sq_off[w] + i*L_w + t <-> (query start+i, key start+t)."""
    counts = np.asarray(counts, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)])
    sq_offsets = np.concatenate([[0], np.cumsum(counts ** 2)])
    N, M = int(offsets[-1]), int(sq_offsets[-1])
    index_0_offsets = np.zeros(N + 1, dtype=np.int64)
    index_1_offsets = np.zeros(N, dtype=np.int64)
    index_0 = np.zeros(M, dtype=np.int64)
    index_1 = np.zeros(M, dtype=np.int64)
    for w, L in enumerate(counts.tolist()):
        start, sv = int(offsets[w]), int(sq_offsets[w])
        t = np.arange(L)
        index_0_offsets[start + t] = sv + L * t
        index_1_offsets[start + t] = sv + t
        ii, tt = np.meshgrid(np.arange(L), np.arange(L), indexing='ij')
        index_0[sv + ii * L + tt] = start + ii
        index_1[sv + ii * L + tt] = start + tt
    index_0_offsets[N] = M
    return index_0_offsets, index_1_offsets, index_0, index_1


def get_indices_params(xyz: torch.Tensor, batch: torch.Tensor, window_size):
    """sptr/utils.py:49-78 with shift_win=False."""
    cluster = grid_cluster(xyz, batch, window_size, None)
    _, v2p, counts = torch.unique(cluster, sorted=True, return_inverse=True, return_counts=True)
    v2p_sorted, sort_idx = torch.sort(v2p, stable=True)
    i0o, i1o, i0, i1 = precompute_all(counts.numpy())
    n_max = int(counts.max())
    return (torch.from_numpy(i0), torch.from_numpy(i0o), n_max, torch.from_numpy(i1), torch.from_numpy(i1o), sort_idx)


def cart2sphere(xyz):  # core/models/sphereformer/spherical_transformer.py:31-36
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    theta = (torch.atan2(y, x) + np.pi) * 180 / np.pi
    beta = torch.atan2(torch.sqrt(x ** 2 + y ** 2), z) * 180 / np.pi
    r = torch.sqrt(x ** 2 + y ** 2 + z ** 2)
    return torch.stack([theta, beta, r], -1)


def exponential_split(xyz, index_0, index_1, relative_position_index, a=0.05 * 0.25, _log_ulps=0):
    """spherical_transformer.py:39-64.  ``_log_ulps`` (fixture generation only): move the logarithm by that many units
    in the last place before the floor -- the CPU and GPU libm differ there; tests/golden/make_golden.py uses it to pick
    scenes without a pair on a radial bin edge."""
    r = xyz[:, 2]
    rel_pos = r[index_0.long()] - r[index_1.long()]
    rel_pos_abs = rel_pos.abs()
    flag_float = (rel_pos >= 0).float()
    lg = torch.log((rel_pos_abs + 2 * a) / a)
    for _ in range(abs(_log_ulps)):
        lg = torch.nextafter(lg, torch.full_like(lg, float('inf') if _log_ulps > 0 else -float('inf')))
    idx = 2 * torch.floor(lg / np.log(2)) - 2
    idx = idx + ((3 * (2 ** (idx // 2)) - 2) * a <= rel_pos_abs).float()
    idx = idx * (2 * flag_float - 1) + (flag_float - 1)
    relative_position_index[:, 2] = idx.long() + 24
    return relative_position_index


def relative_position_index(xyz_ctg, index_0, index_1, window_size, quant_size, quant_grid_length, split_a=None):
    """sptr/modules.py:36-51 (shift_win False): int32 [M,3]."""
    window_size = torch.as_tensor(np.asarray(window_size)).float()
    quant = torch.as_tensor(np.asarray(quant_size)).float()
    xyz_quant = (xyz_ctg - xyz_ctg.min(0)[0] + 0.0) % window_size
    xyz_quant = torch.div(xyz_quant, quant, rounding_mode='floor')
    rel = xyz_quant[index_0.long()] - xyz_quant[index_1.long()]
    rpi = rel + quant_grid_length - 1
    if split_a is not None:
        rpi = exponential_split(xyz_ctg, index_0, index_1, rpi.clone(), a=split_a)
        rpi = torch.clamp(rpi, 0, 2 * quant_grid_length - 1)
    return rpi.int()


def _segment_softmax(src, indptr):
    """scatter_softmax_csr (sptr/utils.py:80-95): per CSR row max-subtracted softmax over dim 0."""
    counts = (indptr[1:] - indptr[:-1]).long()
    seg = torch.repeat_interleave(torch.arange(len(counts)), counts)
    mx = torch.full((len(counts),) + src.shape[1:], -float('inf'), dtype=src.dtype)
    mx = mx.scatter_reduce(0, seg.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce='amax')
    e = (src - mx[seg]).exp()
    s = torch.zeros_like(mx).index_add_(0, seg, e)
    return e / s[seg]


def sparse_self_attention(query, key, value, xyz, index_0, index_0_offsets, n_max, index_1, index_1_offsets, sort_idx,
                          window_size, quant_size, quant_grid_length, table_q, table_k, table_v, split_a=None):
    """sptr/modules.py:11-66, pe_type='contextual', rel_query=rel_key=rel_value=True.
    query/key/value [N,h,d] (query pre-scaled), tables [L,3,h,d] -> [N,h,d]."""
    q, k, v, xyz_ctg = query[sort_idx], key[sort_idx], value[sort_idx], xyz[sort_idx]
    rpi = relative_position_index(xyz_ctg, index_0, index_1, window_size, quant_size, quant_grid_length, split_a).long()
    i0, i1 = index_0.long(), index_1.long()
    ax = torch.arange(3)

    def tsum(table):   # [M,h,d] = T[r1,0] + T[r2,1] + T[r3,2]
        return table[rpi, ax].sum(1)

    attn = (q[i0] * (k[i1] + tsum(table_q))).sum(-1) + (k[i1] * tsum(table_k)).sum(-1)     # [M,h]
    p = _segment_softmax(attn, index_0_offsets.long())
    x = torch.zeros_like(q).index_add_(0, i0, p.unsqueeze(-1) * (v[i1] + tsum(table_v)))
    out = torch.empty_like(x)
    out[sort_idx] = x
    return out


def dense_window_attention(query, key, value, xyz, batch, window_size, quant_size, quant_grid_length,
                           table_q, table_k, table_v, split_a=None):
    """Independent pin: brute-force per-window dense attention with torch.softmax (no CSR, no
    pair lists).  Windows = equal cluster key.  Same contextual relative position terms."""
    cluster = grid_cluster(xyz, batch, window_size, None)
    window_size_t = torch.as_tensor(np.asarray(window_size)).float()
    quant = torch.as_tensor(np.asarray(quant_size)).float()
    qc = torch.div((xyz - xyz.min(0)[0]) % window_size_t, quant, rounding_mode='floor')
    out = torch.zeros_like(query)
    for c in torch.unique(cluster).tolist():
        tok = (cluster == c).nonzero().squeeze(1)
        L = len(tok)
        rel = qc[tok][:, None, :] - qc[tok][None, :, :] + quant_grid_length - 1      # [L,L,3] (query i, key j)
        if split_a is not None:
            r = xyz[tok, 2]
            d = r[:, None] - r[None, :]
            da = d.abs()
            flag = (d >= 0).float()
            idx = 2 * torch.floor(torch.log((da + 2 * split_a) / split_a) / np.log(2)) - 2
            idx = idx + ((3 * (2 ** (idx // 2)) - 2) * split_a <= da).float()
            idx = idx * (2 * flag - 1) + (flag - 1)
            rel[..., 2] = idx.long() + 24
            rel = torch.clamp(rel, 0, 2 * quant_grid_length - 1)
        rel = rel.long()
        ax = torch.arange(3)
        tq = table_q[rel, ax].sum(2)      # [L,L,h,d]
        tk = table_k[rel, ax].sum(2)
        tv = table_v[rel, ax].sum(2)
        q, k, v = query[tok], key[tok], value[tok]
        s = torch.einsum('ihd,jhd->ijh', q, k) + (q[:, None] * tq).sum(-1) + (k[None, :] * tk).sum(-1)
        p = torch.softmax(s, dim=1)
        out[tok] = torch.einsum('ijh,ijhd->ihd', p, v[None, :] + tv)
    return out
