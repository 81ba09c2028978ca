"""CPU restatement of the ten `sptr_cuda` operator kernels, launcher by launcher, in the layouts the
launchers consume (TEST INFRASTRUCTURE: only tests/ may import this).

Citations relative to /root/reference/third_party/SparseTransformer/src/sptr:
  precompute_all                            precompute/precompute_cuda_kernel.cu:4-22
  attention_step1_forward / _backward       attention/attention_cuda_kernel.cu:4-19, 29-62
  attention_step2_forward / _backward       attention/attention_cuda_kernel.cu:77-99, 114-152
  dot_prod_with_idx_forward / _backward     rpe/relative_pos_encoding_cuda_kernel.cu:4-27, 42-99
  dot_prod_with_idx_all_forward             rpe/relative_pos_encoding_cuda_kernel.cu:116-139
  attention_step2_with_rel_pos_value_*      rpe/relative_pos_encoding_cuda_kernel.cu:151-174, 187-254
Pinned (tests/test_oracle_sptr.py): precompute_all on the reference's known-answer fixture
test/test_precompute_all.py:9-42; the decomposition identity of test/test_relative_pos_encoding_op_step1_all.py:87-89
(dot_prod_with_idx_all == dot_prod_with_idx + attention_step1); every backward against torch autograd of its forward.
The sums run over pairs in ascending m (the kernels' loop order), in fp32.
"""
from __future__ import annotations

import torch


def precompute_all(N, n, n_max, counts, offsets, sq_offsets):
    """-> index_0_offsets [N], index_1_offsets [N], index_0 [M], index_1 [M] (int32), as the kernel writes them."""
    from .sptr_ref import precompute_all as _pa
    i0o, i1o, i0, i1 = _pa(counts.numpy())
    return (torch.from_numpy(i0o[:N]).int(), torch.from_numpy(i1o).int(), torch.from_numpy(i0).int(),
            torch.from_numpy(i1).int())


def attention_step1_forward(q_t, k_t, index0, index1):
    """q_t, k_t [h,d,N] -> attn [h,M]"""
    return (q_t[:, :, index0.long()] * k_t[:, :, index1.long()]).sum(1)


def attention_step1_backward(grad_out, index0, index1, q, k):
    """grad_out [M,h]; q, k [N,h,d] -> grad_q, grad_k [N,h,d]"""
    i0, i1 = index0.long(), index1.long()
    gq = torch.zeros_like(q).index_add_(0, i0, grad_out.unsqueeze(-1) * k[i1])
    gk = torch.zeros_like(k).index_add_(0, i1, grad_out.unsqueeze(-1) * q[i0])
    return gq, gk


def attention_step2_forward(attn, v, index0, index1):
    """attn [M,h], v [N,h,d] -> [N,h,d]"""
    return torch.zeros_like(v).index_add_(0, index0.long(), attn.unsqueeze(-1) * v[index1.long()])


def attention_step2_backward(grad_out, index0, index1, attn, v_t):
    """grad_out [N,h,d], attn [M,h], v_t [h,d,N] -> grad_attn [M,h], grad_v [N,h,d]"""
    i0, i1 = index0.long(), index1.long()
    v = v_t.permute(2, 0, 1)
    grad_attn = (grad_out[i0] * v[i1]).sum(-1)
    grad_v = torch.zeros_like(grad_out).index_add_(0, i1, attn.unsqueeze(-1) * grad_out[i0])
    return grad_attn, grad_v


def _tsum_t(table_t, rel_t):
    """table_t [h,d,3,L], rel_t [3,M] -> [h,d,M] = T[..,0,r1] + T[..,1,r2] + T[..,2,r3]"""
    r = rel_t.long()
    return table_t[:, :, 0, r[0]] + table_t[:, :, 1, r[1]] + table_t[:, :, 2, r[2]]


def _tsum(table, rel):
    """table [L,3,h,d], rel [M,3] -> [M,h,d]"""
    r = rel.long()
    return table[r[:, 0], 0] + table[r[:, 1], 1] + table[r[:, 2], 2]


def dot_prod_with_idx_forward(q_t, index_q, k_t, index_k, table_q_t, table_k_t, rel_t):
    """-> [h,M] = q.Tq + k.Tk"""
    return ((q_t[:, :, index_q.long()] * _tsum_t(table_q_t, rel_t)).sum(1)
            + (k_t[:, :, index_k.long()] * _tsum_t(table_k_t, rel_t)).sum(1))


def dot_prod_with_idx_all_forward(q_t, index_q, k_t, index_k, table_q_t, table_k_t, rel_t):
    """-> [h,M] = q.(k + Tq) + k.Tk"""
    qs, ks = q_t[:, :, index_q.long()], k_t[:, :, index_k.long()]
    return (qs * (ks + _tsum_t(table_q_t, rel_t)) + ks * _tsum_t(table_k_t, rel_t)).sum(1)


def dot_prod_with_idx_backward(grad_out, q, index_q, k, index_k, table_q, table_k, rel):
    """grad_out [M,h]; q,k [N,h,d]; tables [L,3,h,d]; rel [M,3] -> grad_q, grad_k, grad_table_q, grad_table_k"""
    iq, ik, r = index_q.long(), index_k.long(), rel.long()
    go = grad_out.unsqueeze(-1)
    gq = torch.zeros_like(q).index_add_(0, iq, _tsum(table_q, rel) * go)
    gk = torch.zeros_like(k).index_add_(0, ik, _tsum(table_k, rel) * go)
    gtq, gtk = torch.zeros_like(table_q), torch.zeros_like(table_k)
    for ax in range(3):
        gtq[:, ax].index_add_(0, r[:, ax], q[iq] * go)
        gtk[:, ax].index_add_(0, r[:, ax], k[ik] * go)
    return gq, gk, gtq, gtk


def attention_step2_with_rel_pos_value_forward(attn, v, index0, index1, table, rel):
    """attn [M,h], v [N,h,d], table [L,3,h,d], rel [M,3] -> [N,h,d]"""
    return torch.zeros_like(v).index_add_(0, index0.long(), attn.unsqueeze(-1) * (v[index1.long()] + _tsum(table, rel)))


def attention_step2_with_rel_pos_value_backward(grad_out, index0, index1, attn, v_t, table_t, rel_t):
    """grad_out [N,h,d], attn [M,h], v_t [h,d,N], table_t [h,d,3,L], rel_t [3,M]
    -> grad_attn [M,h], grad_v [N,h,d], grad_table [L,3,h,d]"""
    i0, i1, r = index0.long(), index1.long(), rel_t.long()
    v = v_t.permute(2, 0, 1)
    tv = _tsum_t(table_t, rel_t).permute(2, 0, 1)            # [M,h,d]
    grad_attn = (grad_out[i0] * (v[i1] + tv)).sum(-1)
    g = attn.unsqueeze(-1) * grad_out[i0]                    # [M,h,d]
    grad_v = torch.zeros_like(grad_out).index_add_(0, i1, g)
    L = table_t.shape[-1]
    grad_table = torch.zeros(L, 3, *grad_out.shape[1:], dtype=grad_out.dtype)
    for ax in range(3):
        grad_table[:, ax].index_add_(0, r[ax], g)
    return grad_attn, grad_v, grad_table
