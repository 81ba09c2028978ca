from ._ntuple import make_ntuple

__all__ = ['make_ntuple']
