from ...ts_ref import make_ntuple
from . import quantize, collate

__all__ = ['make_ntuple', 'quantize', 'collate']
