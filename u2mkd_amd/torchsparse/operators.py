"""torchsparse.cat (core/models/semantickitti/spvcnn.py:116,120,128,132)."""
import torch

from .tensor import SparseTensor

__all__ = ['cat']


def cat(inputs):
    feats = torch.cat([t.feats for t in inputs], dim=1)
    output = SparseTensor(coords=inputs[0].coords, feats=feats, stride=inputs[0].stride)
    output.cmaps = inputs[0].cmaps
    output.kmaps = inputs[0].kmaps
    return output
