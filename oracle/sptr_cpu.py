"""CPU drop-in for ``third_party.SparseTransformer.sptr`` over oracle.sptr_ref
(TEST INFRASTRUCTURE): the four names core/models/sphereformer/spherical_transformer.py:7
imports, with the reference's signatures (sptr/modules.py:11-33, sptr/utils.py:9-17,49-78,
sptr/__init__.py:4-32), so the reference's own SphereFormer module runs on CPU in the
build container for golden generation."""
import numbers

import numpy as np
import torch

from . import sptr_ref as S

__all__ = ['to_3d_numpy', 'SparseTrTensor', 'sparse_self_attention', 'get_indices_params']


def to_3d_numpy(size):
    if isinstance(size, numbers.Number):
        size = np.array([size, size, size]).astype(np.float32)
    elif isinstance(size, list):
        size = np.array(size)
    elif isinstance(size, np.ndarray):
        size = size
    else:
        raise ValueError("size is either a number, or a list, or a np.ndarray")
    return size


class SparseTrTensor(object):
    def __init__(self, query_feats, query_indices, spatial_shape, batch_size, key_feats=None, value_feats=None,
                 key_indices=None):
        self.query_feats = query_feats
        self.key_feats = key_feats
        self.value_feats = value_feats
        self.query_indices = query_indices
        self.key_indices = key_indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}

    def find_indice_params(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)


def get_indices_params(xyz, batch, window_size, shift_win: bool):
    assert not shift_win
    return S.get_indices_params(xyz.detach(), batch, np.asarray(window_size))


def sparse_self_attention(query, key, value, xyz, index_0, index_0_offsets, n_max, index_1, index_1_offsets, sort_idx,
                          window_size, shift_win, pe_type='none', rel_query=False, rel_key=False, rel_value=False,
                          quant_size=None, quant_grid_length=None, relative_pos_query_table=None,
                          relative_pos_key_table=None, relative_pos_value_table=None, split_func=None):
    assert pe_type == 'contextual' and rel_query and rel_key and rel_value and not shift_win
    a = None
    if split_func is not None:
        a = split_func.keywords.get('a', 0.05 * 0.25)
    return S.sparse_self_attention(query, key, value, xyz, index_0, index_0_offsets, n_max, index_1, index_1_offsets,
                                   sort_idx, np.asarray(window_size), np.asarray(quant_size), quant_grid_length,
                                   relative_pos_query_table, relative_pos_key_table, relative_pos_value_table, a)
