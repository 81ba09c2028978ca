"""torchsparse.nn.utils (core/models/utils.py:5-7,84,141)."""
import numpy as np
import torch

from ..tensor import SparseTensor
from ..utils import make_ntuple

__all__ = ['get_kernel_offsets', 'fapply']

_OFFSET_CACHE = {}


def get_kernel_offsets(size, stride=1, dilation=1, device='cpu'):
    """int32 [K,3].  Odd kernel volume: x fastest; even: z fastest (v1.4.0,
    SURVEY.md Appendix A-3).  Cached per (size, stride, dilation, device): the
    table is tiny host arithmetic, the cache removes a host->device copy per call."""
    size = make_ntuple(size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    key = (size, stride, dilation, str(device))
    hit = _OFFSET_CACHE.get(key)
    if hit is not None:
        return hit
    offsets = [(np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k]) for k in range(3)]
    if np.prod(size) % 2 == 1:
        offsets = [[x, y, z] for z in offsets[2] for y in offsets[1] for x in offsets[0]]
    else:
        offsets = [[x, y, z] for x in offsets[0] for y in offsets[1] for z in offsets[2]]
    out = torch.tensor(offsets, dtype=torch.int, device=device)
    _OFFSET_CACHE[key] = out
    return out


def fapply(input, fn, *args, **kwargs):
    feats = fn(input.feats, *args, **kwargs)
    output = SparseTensor(coords=input.coords, feats=feats, stride=input.stride)
    output.cmaps = input.cmaps
    output.kmaps = input.kmaps
    return output
