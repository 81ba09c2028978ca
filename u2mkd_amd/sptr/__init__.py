"""Drop-in for ``third_party.SparseTransformer.sptr`` (the four names
core/models/sphereformer/spherical_transformer.py:7 imports) on MI355X.

``get_indices_params`` returns a :class:`WindowPlan` in place of the reference's
``index_0`` (and inert placeholders for the other M-sized index tensors): the
HIP attention never materialises the sum_w L_w^2 pair lists, it walks windows
of sorted tokens.  ``sparse_self_attention`` keeps the reference signature
(sptr/modules.py:11-33)."""
import numbers

import numpy as np

from .functional import WindowPlan, get_indices_params, sparse_self_attention, window_attention

__all__ = ['to_3d_numpy', 'SparseTrTensor', 'sparse_self_attention', 'get_indices_params', 'WindowPlan',
           'window_attention']


def to_3d_numpy(size):
    """sptr/utils.py:9-17 (ndarray inputs are returned as the SAME object -- the aliasing of
    SURVEY Appendix C-1 depends on it)."""
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
    """sptr/__init__.py:4-32."""

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

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_params(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)
