from . import functional, utils
from .modules import Conv3d, BatchNorm, ReLU, LeakyReLU

__all__ = ['functional', 'utils', 'Conv3d', 'BatchNorm', 'ReLU', 'LeakyReLU']
