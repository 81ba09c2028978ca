"""Drop-in for ``third_party.SparseTransformer.sptr`` on MI355X: the four names
core/models/sphereformer/spherical_transformer.py:7 imports (``to_3d_numpy``, ``SparseTrTensor``,
``sparse_self_attention``, ``get_indices_params``).

``get_indices_params`` returns a :class:`WindowPlan` in place of the reference's ``index_0`` (and inert
placeholders for the other M-sized index tensors): the HIP attention never materialises the
sum_w L_w^2 pair lists, it walks windows of sorted tokens.  ``sparse_self_attention`` keeps the
reference signature (sptr/modules.py:11-33).  ``u2mkd_amd.install_as_sptr()`` registers this package
under the reference's import path."""
import numbers

import numpy as np

from .functional import WindowPlan, get_indices_params, packed_window_attention, sparse_self_attention, window_attention

__all__ = ['to_3d_numpy', 'SparseTrTensor', 'sparse_self_attention', 'get_indices_params', 'WindowPlan',
           'window_attention', 'packed_window_attention']


def to_3d_numpy(size):
    """A window / quantisation size as a length-3 array (contract of sptr/utils.py:9-17).

    A scalar becomes a float32 triple, a list becomes a NEW array, an ndarray is handed back AS THE SAME
    OBJECT -- callers mutate it in place afterwards and rely on the aliasing (SURVEY.md Appendix C-1:
    all four SphereFormer blocks end up sharing one ``quant_size_sphere``)."""
    if isinstance(size, np.ndarray):
        return size
    if isinstance(size, numbers.Number):
        return np.full(3, size, dtype=np.float32)
    if isinstance(size, list):
        return np.asarray(size).copy()
    raise ValueError("size is either a number, or a list, or a np.ndarray")


class SparseTrTensor:
    """Plain record of the token tensors of one attention call plus a per-tensor cache of index
    structures, the attribute surface of the reference's class (sptr/__init__.py:4-32): query / key /
    value features, their integer coordinates (batch index first), the grid extent and the batch size."""

    __slots__ = ('query_feats', 'key_feats', 'value_feats', 'query_indices', 'key_indices', 'spatial_shape',
                 'batch_size', 'indice_dict')

    def __init__(self, query_feats, query_indices, spatial_shape, batch_size, key_feats=None, value_feats=None,
                 key_indices=None):
        self.query_feats, self.query_indices = query_feats, query_indices
        self.key_feats, self.value_feats, self.key_indices = key_feats, value_feats, key_indices
        self.spatial_shape, self.batch_size = spatial_shape, batch_size
        self.indice_dict = {}

    @property
    def spatial_size(self):
        """Number of cells of the grid."""
        return int(np.prod(self.spatial_shape))

    def find_indice_params(self, key):
        """Cached index structure stored under ``key`` (None when absent or when ``key`` is None)."""
        return None if key is None else self.indice_dict.get(key)
