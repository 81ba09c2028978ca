"""torchsparse.nn.utils: get_kernel_offsets, fapply (core/models/utils.py:5-7,84,141)."""
import torch

from ...ts_ref import get_kernel_offsets as _gko
from ..tensor import SparseTensor

__all__ = ['get_kernel_offsets', 'fapply']


def get_kernel_offsets(size, stride=1, dilation=1, device='cpu'):
    return torch.from_numpy(_gko(size, stride, dilation))


def fapply(input, fn, *args, **kwargs):
    feats = fn(input.feats, *args, **kwargs)
    output = SparseTensor(coords=input.coords, feats=feats, stride=input.stride)
    output.cmaps = input.cmaps
    output.kmaps = input.kmaps
    return output
