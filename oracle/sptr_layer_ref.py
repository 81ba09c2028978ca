"""Restatement of sptr's PYTHON LAYER over a ``sptr_cuda``-shaped backend (TEST INFRASTRUCTURE: only tests/ may import
this; row a11 / boundary 2 of SURVEY.md section 8).

Follows, line for line, relative to /root/reference/third_party/SparseTransformer:
  * sptr/functional.py:9-82     AttentionStep1          (not on the model's path: pe_type 'none')
  * sptr/functional.py:84-144   AttentionStep2          (likewise)
  * sptr/functional.py:146-167  PrecomputeAll
  * sptr/functional.py:248-333  DotProdWithIdxAll       (backward = dot_prod_with_idx_backward + attention_step1_backward)
  * sptr/functional.py:335-405  AttentionStep2WithRelPosValue
  * sptr/utils.py:49-78         get_indices_params      (voxel_grid = oracle.sptr_ref.grid_cluster: torch_geometric is an
                                                        un-vendored dependency)
  * sptr/utils.py:80-95         scatter_softmax_csr     (torch_scatter segment_csr / gather_csr restated with torch ops)
  * sptr/modules.py:11-66       sparse_self_attention
``layer(backend)`` binds the restatement to a backend module exposing the ten functions of
src/sptr/pointops_api.cpp:9-20 with the pybind signatures (outputs pre-zeroed and passed in, ``None`` returned):
  * ``CpuBackend()`` below = oracle.sptr_ops_ref behind those signatures (pins this file on the CPU against
    oracle.sptr_ref.sparse_self_attention, tests/test_oracle_sptr.py);
  * ``u2mkd_amd.sptr.sptr_cuda`` on the GPU box = THE REFERENCE'S CALL SEQUENCE over the product's ten C-ABI entries, held
    against the fused kernel the product's model runs (tests/test_gpu_reference_sequence.py).
The reference allocates with ``torch.cuda.FloatTensor(..).zero_()``; here ``torch.zeros(.., device=<input's>)``.
"""
from __future__ import annotations

import types

import numpy as np
import torch
from torch.autograd import Function

from . import sptr_ops_ref as K
from .sptr_ref import grid_cluster


class CpuBackend:
    """oracle.sptr_ops_ref behind the ``sptr_cuda`` signatures (results copied into the pre-zeroed outputs)."""

    @staticmethod
    def precompute_all_cuda(N, n, n_max, counts, offsets, sq_offsets, index_0_offsets, index_1_offsets, index_0, index_1):
        a, b, c, d = K.precompute_all(N, n, n_max, counts.cpu(), offsets.cpu(), sq_offsets.cpu())
        index_0_offsets.copy_(a), index_1_offsets.copy_(b), index_0.copy_(c), index_1.copy_(d)

    @staticmethod
    def attention_step1_forward_cuda(N_q, N_k, M, h, hdim, n_max, q, k, index0, index1, output):
        output.copy_(K.attention_step1_forward(q, k, index0, index1))

    @staticmethod
    def attention_step1_backward_cuda(N, M, h, hdim, n_max, grad_out, index0, index0_offsets, index1, index1_offsets, q, k,
                                      grad_q, grad_k):
        a, b = K.attention_step1_backward(grad_out, index0, index1, q, k)
        grad_q.copy_(a), grad_k.copy_(b)

    @staticmethod
    def attention_step2_forward_cuda(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, output):
        index0 = torch.repeat_interleave(torch.arange(N), torch.diff(index0_offsets.long()))
        output.copy_(K.attention_step2_forward(attn, v, index0, index1))

    @staticmethod
    def attention_step2_backward_cuda(N, M, h, hdim, n_max, grad_out, index0, index0_offsets, index1, index1_offsets, attn, v,
                                      grad_attn, grad_v):
        a, b = K.attention_step2_backward(grad_out, index0, index1, attn, v)
        grad_attn.copy_(a), grad_v.copy_(b)

    @staticmethod
    def dot_prod_with_idx_forward_cuda(N, M, h, hdim, n_max, L, q, index_q, index_q_offsets, k, index_k, table_q, table_k,
                                       rel_idx, output):
        output.copy_(K.dot_prod_with_idx_forward(q, index_q, k, index_k, table_q, table_k, rel_idx))

    @staticmethod
    def dot_prod_with_idx_all_forward_cuda(N, M, h, hdim, n_max, L, q, index_q, index_q_offsets, k, index_k, table_q, table_k,
                                           rel_idx, output):
        output.copy_(K.dot_prod_with_idx_all_forward(q, index_q, k, index_k, table_q, table_k, rel_idx))

    @staticmethod
    def dot_prod_with_idx_backward_cuda(N, M, h, hdim, n_max, L, grad_out, q, index_q_offsets, k, index_k_offsets, index_k,
                                        table_q, table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k):
        index_q = torch.repeat_interleave(torch.arange(N), torch.diff(index_q_offsets.long()))
        a, b, c, d = K.dot_prod_with_idx_backward(grad_out, q, index_q, k, index_k, table_q, table_k, rel_idx)
        grad_q.copy_(a), grad_k.copy_(b), grad_table_q.copy_(c), grad_table_k.copy_(d)

    @staticmethod
    def attention_step2_with_rel_pos_value_forward_cuda(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, table, rel_idx,
                                                        output):
        index0 = torch.repeat_interleave(torch.arange(N), torch.diff(index0_offsets.long()))
        output.copy_(K.attention_step2_with_rel_pos_value_forward(attn, v, index0, index1, table, rel_idx))

    @staticmethod
    def attention_step2_with_rel_pos_value_backward_cuda(N, M, h, hdim, L, n_max, grad_out, index0, index0_offsets, index1,
                                                         index1_offsets, attn, v, table, rel_idx, grad_attn, grad_v,
                                                         grad_table):
        a, b, c = K.attention_step2_with_rel_pos_value_backward(grad_out, index0, index1, attn, v, table, rel_idx)
        grad_attn.copy_(a), grad_v.copy_(b), grad_table.copy_(c)


def scatter_softmax_csr(src: torch.Tensor, indptr: torch.Tensor, dim: int = 0):
    """sptr/utils.py:80-95: per CSR segment, exp(src - max) / sum -- segment_csr(reduce='max'/'sum') and gather_csr restated
    with index_reduce / index_add over the segment id of every row (differentiable like torch_scatter's)."""
    n_seg = indptr.shape[0] - 1
    seg = torch.repeat_interleave(torch.arange(n_seg, device=src.device), torch.diff(indptr))
    mx = torch.full((n_seg,) + src.shape[1:], float('-inf'), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, seg.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src.detach(), reduce='amax', include_self=True)
    recentered_scores_exp = (src - mx[seg]).exp()
    sum_per_index = torch.zeros((n_seg,) + src.shape[1:], dtype=src.dtype, device=src.device).index_add(0, seg, recentered_scores_exp)
    return recentered_scores_exp.div(sum_per_index[seg])


def layer(sptr_cuda):
    """The Python layer bound to ``sptr_cuda`` (a module or object with the ten ``*_cuda`` functions): a namespace with
    ``precompute_all, attention_step1, attention_step2, dot_prod_with_idx_all, attention_step2_with_rel_pos_value,
    get_indices_params, sparse_self_attention``."""

    class AttentionStep1(Function):                   # functional.py:9-82
        @staticmethod
        def forward(ctx, q, k, index0, index0_offsets, index1, index1_offsets, n_max):
            assert q.is_contiguous() and k.is_contiguous() and index0_offsets.is_contiguous() and index1.is_contiguous()
            N_q, h, hdim = q.shape
            N_k = k.shape[0]
            M = index1.shape[0]
            output = torch.zeros(h, M, dtype=torch.float32, device=q.device)
            q_transpose = q.permute(1, 2, 0).contiguous()
            k_transpose = k.permute(1, 2, 0).contiguous()
            sptr_cuda.attention_step1_forward_cuda(N_q, N_k, M, h, hdim, n_max, q_transpose, k_transpose, index0, index1, output)
            output = output.permute(1, 0).contiguous()
            ctx.N_q, ctx.N_k, ctx.n_max = N_q, N_k, n_max
            ctx.save_for_backward(q, k, index0, index0_offsets, index1, index1_offsets)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            q, k, index0, index0_offsets, index1, index1_offsets = ctx.saved_tensors
            M, h = grad_output.shape
            hdim = q.shape[2]
            assert 512 % hdim == 0
            grad_output = grad_output.contiguous()
            grad_q = torch.zeros(ctx.N_q, h, hdim, dtype=torch.float32, device=q.device)
            grad_k = torch.zeros(ctx.N_k, h, hdim, dtype=torch.float32, device=q.device)
            sptr_cuda.attention_step1_backward_cuda(ctx.N_q, M, h, hdim, ctx.n_max, grad_output, index0, index0_offsets, index1,
                                                    index1_offsets, q, k, grad_q, grad_k)
            return grad_q, grad_k, None, None, None, None, None

    class AttentionStep2(Function):                   # functional.py:84-144
        @staticmethod
        def forward(ctx, attn, v, index0, index0_offsets, index1, index1_offsets, n_max):
            assert attn.is_contiguous() and v.is_contiguous() and index0_offsets.is_contiguous() and index1.is_contiguous()
            M, h = attn.shape
            _, h, hdim = v.shape
            N = index0_offsets.shape[0] - 1
            output = torch.zeros(N, h, hdim, dtype=torch.float32, device=v.device)
            assert attn.shape[1] == h and 512 % hdim == 0
            sptr_cuda.attention_step2_forward_cuda(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, output)
            ctx.n_max = n_max
            ctx.save_for_backward(attn, v, index0, index0_offsets, index1, index1_offsets)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            attn, v, index0, index0_offsets, index1, index1_offsets = ctx.saved_tensors
            N, h, hdim = grad_output.shape
            N_k, M = v.shape[0], attn.shape[0]
            grad_output = grad_output.contiguous()
            grad_attn = torch.zeros(M, h, dtype=torch.float32, device=v.device)
            grad_v = torch.zeros(N_k, h, hdim, dtype=torch.float32, device=v.device)
            v = v.permute(1, 2, 0).contiguous()
            sptr_cuda.attention_step2_backward_cuda(N, M, h, hdim, ctx.n_max, grad_output, index0, index0_offsets, index1,
                                                    index1_offsets, attn, v, grad_attn, grad_v)
            return grad_attn, grad_v, None, None, None, None, None

    def precompute_all(N, n, n_max, counts):          # functional.py:146-167
        assert counts.is_contiguous()
        offsets = torch.cat([counts.new_zeros(1), counts.cumsum(-1)], 0)
        sq_offsets = torch.cat([counts.new_zeros(1), (counts ** 2).cumsum(-1)], 0)
        M = sq_offsets[-1].item()
        dev = counts.device
        index_0_offsets = torch.zeros(N, dtype=torch.int32, device=dev)
        index_1_offsets = torch.zeros(N, dtype=torch.int32, device=dev)
        index_0 = torch.zeros(M, dtype=torch.int32, device=dev)
        index_1 = torch.zeros(M, dtype=torch.int32, device=dev)
        sptr_cuda.precompute_all_cuda(N, n, n_max, counts.int(), offsets.int(), sq_offsets.int(), index_0_offsets, index_1_offsets,
                                      index_0, index_1)
        index_0_offsets = torch.cat([index_0_offsets, torch.tensor([M], device=dev)], 0)       # (int64 from here, as in the reference)
        return index_0_offsets, index_1_offsets, index_0, index_1

    class DotProdWithIdxAll(Function):                # functional.py:248-333
        @staticmethod
        def forward(ctx, q, index_q, index_q_offsets, k, index_k, index_k_offsets, table_q, table_k, rel_idx, n_max):
            assert q.is_contiguous() and index_q.is_contiguous() and index_q_offsets.is_contiguous() and k.is_contiguous() \
                and index_k.is_contiguous() and table_q.is_contiguous() and table_k.is_contiguous() and rel_idx.is_contiguous()
            N, h, hdim = q.shape
            M = index_k.shape[0]
            L = table_q.shape[0]
            assert table_k.shape[0] == L and q.shape[0] == k.shape[0]
            assert L > rel_idx.max(), 'L = {}, while rel_idx.max() = {}'.format(L, rel_idx.max())
            assert L <= 50
            output = torch.zeros(h, M, dtype=torch.float32, device=q.device)
            q_transpose = q.permute(1, 2, 0).contiguous()
            k_transpose = k.permute(1, 2, 0).contiguous()
            table_q_transpose = table_q.permute(2, 3, 1, 0).contiguous()
            table_k_transpose = table_k.permute(2, 3, 1, 0).contiguous()
            rel_idx_transpose = rel_idx.permute(1, 0).contiguous()
            sptr_cuda.dot_prod_with_idx_all_forward_cuda(N, M, h, hdim, n_max, L, q_transpose, index_q, index_q_offsets, k_transpose,
                                                         index_k, table_q_transpose, table_k_transpose, rel_idx_transpose, output)
            output = output.permute(1, 0).contiguous()
            ctx.n_max = n_max
            ctx.save_for_backward(q, index_q_offsets, index_q, k, index_k_offsets, index_k, table_q, table_k, rel_idx)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            q, index_q_offsets, index_q, k, index_k_offsets, index_k, table_q, table_k, rel_idx = ctx.saved_tensors
            M, h = grad_output.shape
            N, _, hdim = q.shape
            L = table_q.shape[0]
            n_max = ctx.n_max
            N_k = k.shape[0]
            grad_output = grad_output.contiguous()
            assert L <= 50 and 512 % hdim == 0
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=q.device)
            grad_q, grad_table_q, grad_k, grad_table_k = z(N, h, hdim), z(L, 3, h, hdim), z(N_k, h, hdim), z(L, 3, h, hdim)
            sptr_cuda.dot_prod_with_idx_backward_cuda(N, M, h, hdim, n_max, L, grad_output, q, index_q_offsets, k, index_k_offsets,
                                                      index_k, table_q, table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k)
            grad_q_2, grad_k_2 = z(N, h, hdim), z(N, h, hdim)
            sptr_cuda.attention_step1_backward_cuda(N, M, h, hdim, n_max, grad_output, index_q, index_q_offsets, index_k,
                                                    index_k_offsets, q, k, grad_q_2, grad_k_2)
            grad_q += grad_q_2
            grad_k += grad_k_2
            return grad_q, None, None, grad_k, None, None, grad_table_q, grad_table_k, None, None

    class AttentionStep2WithRelPosValue(Function):    # functional.py:335-405
        @staticmethod
        def forward(ctx, attn, v, index0, index0_offsets, n_max, index1, index1_offsets, table, rel_idx):
            assert attn.is_contiguous() and v.is_contiguous() and index0.is_contiguous() and index0_offsets.is_contiguous() \
                and index1.is_contiguous() and index1_offsets.is_contiguous() and table.is_contiguous() and rel_idx.is_contiguous()
            M, h = attn.shape
            _, h, hdim = v.shape
            N = index0_offsets.shape[0] - 1
            L = table.shape[0]
            output = torch.zeros(N, h, hdim, dtype=torch.float32, device=v.device)
            assert hdim == 16 and L <= 50
            sptr_cuda.attention_step2_with_rel_pos_value_forward_cuda(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, table,
                                                                      rel_idx, output)
            ctx.n_max = n_max
            ctx.save_for_backward(attn, v, index0, index0_offsets, index1, index1_offsets, table, rel_idx)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            n_max = ctx.n_max
            attn, v, index0, index0_offsets, index1, index1_offsets, table, rel_idx = ctx.saved_tensors
            N, h, hdim = grad_output.shape
            N_k, M, L = v.shape[0], attn.shape[0], table.shape[0]
            grad_output = grad_output.contiguous()
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=v.device)
            grad_attn, grad_v, grad_table = z(M, h), z(N_k, h, hdim), z(L, 3, h, hdim)
            table = table.permute(2, 3, 1, 0).contiguous()
            v = v.permute(1, 2, 0).contiguous()
            rel_idx = rel_idx.permute(1, 0).contiguous()
            sptr_cuda.attention_step2_with_rel_pos_value_backward_cuda(N, M, h, hdim, L, n_max, grad_output, index0, index0_offsets,
                                                                       index1, index1_offsets, attn, v, table, rel_idx, grad_attn,
                                                                       grad_v, grad_table)
            return grad_attn, grad_v, None, None, None, None, None, grad_table, None

    attention_step1 = AttentionStep1.apply
    attention_step2 = AttentionStep2.apply
    dot_prod_with_idx_all = DotProdWithIdxAll.apply
    attention_step2_with_rel_pos_value = AttentionStep2WithRelPosValue.apply

    def get_indices_params(xyz, batch, window_size, shift_win: bool):      # utils.py:19-78 (shift_win False on the model's path)
        assert not shift_win
        if isinstance(window_size, (list, np.ndarray)):
            window_size = torch.from_numpy(np.asarray(window_size)).type_as(xyz).to(xyz.device)
        else:
            window_size = torch.tensor([window_size] * 3).type_as(xyz).to(xyz.device)
        cluster = grid_cluster(xyz.cpu(), batch.cpu(), window_size.cpu(), None).to(xyz.device)     # voxel_grid (un-vendored)
        unique, v2p_map, counts = torch.unique(cluster, sorted=True, return_inverse=True, return_counts=True)
        k = counts.max().item()
        v2p_map, sort_idx = v2p_map.sort(stable=True)
        n = counts.shape[0]
        N = v2p_map.shape[0]
        n_max = k
        index_0_offsets, index_1_offsets, index_0, index_1 = precompute_all(N, n, n_max, counts)
        return index_0.long(), index_0_offsets, n_max, index_1.long(), index_1_offsets, sort_idx

    def sparse_self_attention(query, key, value, xyz, index_0, index_0_offsets, n_max, index_1, index_1_offsets, sort_idx,
                              window_size, shift_win, pe_type='none', rel_query=False, rel_key=False, rel_value=False,
                              quant_size=None, quant_grid_length=None, relative_pos_query_table=None,
                              relative_pos_key_table=None, relative_pos_value_table=None, split_func=None):   # modules.py:11-66
        query = query[sort_idx]
        key = key[sort_idx]
        value = value[sort_idx]
        xyz_ctg = xyz[sort_idx]
        dev = xyz.device
        if pe_type == 'contextual' and rel_query and rel_key:
            window_size = torch.from_numpy(window_size).float().to(dev)
            shift_size = 1 / 2 * window_size if shift_win else 0.0
            xyz_quant = (xyz_ctg - xyz_ctg.min(0)[0] + shift_size) % window_size
            xyz_quant = torch.div(xyz_quant, torch.from_numpy(quant_size).float().to(dev), rounding_mode='floor')
            relative_position = xyz_quant[index_0.long()] - xyz_quant[index_1.long()]
            relative_position_index = relative_position + quant_grid_length - 1
            if split_func:
                relative_position_index = split_func(xyz_ctg, index_0, index_1, relative_position_index.clone())
                relative_position_index = torch.clamp(relative_position_index, 0, 2 * quant_grid_length - 1)
            relative_position_index = relative_position_index.int()
            attn_flat = dot_prod_with_idx_all(query, index_0, index_0_offsets, key, index_1, index_1_offsets,
                                              relative_pos_query_table, relative_pos_key_table, relative_position_index, n_max)
        else:
            attn_flat = attention_step1(query, key, index_0, index_0_offsets, index_1, index_1_offsets, n_max)
        softmax_attn_flat = scatter_softmax_csr(src=attn_flat, indptr=index_0_offsets.long(), dim=0)
        if pe_type == 'contextual' and rel_value:
            x = attention_step2_with_rel_pos_value(softmax_attn_flat, value, index_0, index_0_offsets, n_max, index_1,
                                                   index_1_offsets, relative_pos_value_table, relative_position_index)
        else:
            x = attention_step2(softmax_attn_flat, value, index_0, index_0_offsets, index_1, index_1_offsets, n_max)
        out = torch.empty_like(x)
        out[sort_idx] = x
        return out

    return types.SimpleNamespace(precompute_all=precompute_all, attention_step1=attention_step1, attention_step2=attention_step2,
                                 dot_prod_with_idx_all=dot_prod_with_idx_all,
                                 attention_step2_with_rel_pos_value=attention_step2_with_rel_pos_value,
                                 get_indices_params=get_indices_params, sparse_self_attention=sparse_self_attention,
                                 scatter_softmax_csr=scatter_softmax_csr, to_3d_numpy=_to_3d_numpy)


def _to_3d_numpy(size):                               # sptr/utils.py:9-17
    import numbers
    if isinstance(size, numbers.Number):
        return np.array([size, size, size]).astype(np.float32)
    if isinstance(size, list):
        return np.array(size)
    if isinstance(size, np.ndarray):
        return size
    raise ValueError('size is either a number, or a list, or a np.ndarray')
