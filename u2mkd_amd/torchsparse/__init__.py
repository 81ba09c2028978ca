"""Drop-in for the torchsparse v1.4.0 API surface used by U2MKD."""
from .tensor import SparseTensor, PointTensor
from .operators import cat
from . import nn, utils

__version__ = '1.4.0'
__all__ = ['SparseTensor', 'PointTensor', 'cat', 'nn', 'utils']
from .utils import quantize, collate  # noqa: E402,F401  (torchsparse.utils.quantize / .collate)
