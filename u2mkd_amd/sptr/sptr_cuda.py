"""Stand-in for the ``sptr_cuda`` extension module of third_party/SparseTransformer (src/sptr/pointops_api.cpp:9-20): the
ten functions its pybind block exports, same names, same argument lists (plain ints, then tensors in the launchers' layouts,
outputs pre-zeroed by the caller, ``None`` returned) -- for a maintainer who keeps sptr's own Python layer
(``sptr/functional.py:5``: ``import sptr_cuda``) and its M-sized pair arrays.  Each call is one C-ABI entry
``u2mkd_sptr_<name>`` (include/u2mkd_hip.h, csrc/sptr_ops.hip) on torch's CURRENT stream (the reference launches on
stream 0).  ``u2mkd_amd.install_as_sptr_cuda()`` registers this module as ``sptr_cuda``.

The product's own model does not come through here: ``u2mkd_amd.sptr`` runs the fused attention
(``u2mkd_sptr_attention_forward / _backward``), which never materialises arrays of size M."""
import torch

from .. import _lib as L

__all__ = ['attention_step1_forward_cuda', 'attention_step1_backward_cuda', 'attention_step2_forward_cuda',
           'attention_step2_backward_cuda', 'precompute_all_cuda', 'dot_prod_with_idx_forward_cuda',
           'dot_prod_with_idx_backward_cuda', 'attention_step2_with_rel_pos_value_forward_cuda',
           'attention_step2_with_rel_pos_value_backward_cuda', 'dot_prod_with_idx_all_forward_cuda']


def _args(name, args):
    out = []
    for a in args:
        if torch.is_tensor(a):
            # what the reference's launchers assume and `data_ptr<float>() / <int>()` enforce (attention_cuda.cpp:11-15)
            if not a.is_cuda:
                raise RuntimeError(f'{name}: tensors must live on the HIP device (no CPU fallback)')
            if a.dtype not in (torch.float32, torch.int32):
                raise RuntimeError(f'{name}: expected a float32 or int32 tensor, got {a.dtype}')
            if not a.is_contiguous():
                raise RuntimeError(f'{name}: tensors must be contiguous (sptr/functional.py asserts it)')
            out.append(a.data_ptr())
        else:
            out.append(int(a))
    return out


def _entry(name):
    def fn(*args):
        L.call('u2mkd_sptr_' + name, *_args(name + '_cuda', args), L.stream())
    fn.__name__ = fn.__qualname__ = name + '_cuda'
    return fn


# attention/attention_cuda.cpp:7-60
attention_step1_forward_cuda = _entry('attention_step1_forward')        # (N_q, N_k, M, h, hdim, n_max, q, k, index0, index1, attn)
attention_step1_backward_cuda = _entry('attention_step1_backward')      # (N, M, h, hdim, n_max, grad_out, index0, index0_offsets, index1, index1_offsets, q, k, grad_q, grad_k)
attention_step2_forward_cuda = _entry('attention_step2_forward')        # (N, M, h, hdim, n_max, attn, v, index0_offsets, index1, output)
attention_step2_backward_cuda = _entry('attention_step2_backward')      # (N, M, h, hdim, n_max, grad_out, index0, index0_offsets, index1, index1_offsets, attn, v, grad_attn, grad_v)
# precompute/precompute.cpp:7-17
precompute_all_cuda = _entry('precompute_all')                          # (N, n, n_max, counts, offsets, sq_offsets, index_0_offsets, index_1_offsets, index_0, index_1)
# rpe/relative_pos_encoding_cuda.cpp
dot_prod_with_idx_forward_cuda = _entry('dot_prod_with_idx_forward')    # (N, M, h, hdim, n_max, L, q, index_q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output)
dot_prod_with_idx_all_forward_cuda = _entry('dot_prod_with_idx_all_forward')
dot_prod_with_idx_backward_cuda = _entry('dot_prod_with_idx_backward')  # (N, M, h, hdim, n_max, L, grad_out, q, index_q_offsets, k, index_k_offsets, index_k, table_q, table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k)
attention_step2_with_rel_pos_value_forward_cuda = _entry('attention_step2_with_rel_pos_value_forward')    # (N, M, h, hdim, n_max, attn, v, index0_offsets, index1, table, rel_idx, output)
attention_step2_with_rel_pos_value_backward_cuda = _entry('attention_step2_with_rel_pos_value_backward')  # (N, M, h, hdim, L, n_max, grad_out, index0, index0_offsets, index1, index1_offsets, attn, v, table, rel_idx, grad_attn, grad_v, grad_table)
