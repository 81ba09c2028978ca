"""CPU drop-in for the torchsparse v1.4.0 Python API (TEST INFRASTRUCTURE).

Lets the reference's own pure-Python model files (core/models/build_blocks.py,
core/models/utils.py, core/models/semantickitti/spvcnn.py, ...) import and run
in the build container over ``oracle.ts_ref`` so golden vectors can be
generated (tests/golden/make_golden.py).  Never imported by ``u2mkd_amd``.

Surface = SURVEY.md §8b Boundary 1 (names grep'd from the reference's call
sites); semantics = torchsparse v1.4.0 (README.md:44-48), restated.
"""
import sys

from .tensor import SparseTensor, PointTensor
from .operators import cat
from . import nn, utils

__version__ = '1.4.0'
__all__ = ['SparseTensor', 'PointTensor', 'cat', 'nn', 'utils', 'install']


def install():
    """Register this package as ``torchsparse`` in sys.modules (tests only)."""
    import importlib
    base = __name__
    sys.modules['torchsparse'] = sys.modules[base]
    for sub in ('tensor', 'operators', 'nn', 'nn.functional', 'nn.utils', 'nn.modules',
                'utils', 'utils.quantize', 'utils.collate'):
        sys.modules['torchsparse.' + sub] = importlib.import_module(base + '.' + sub)
