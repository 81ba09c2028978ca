"""torchsparse.utils.collate (core/datasets/lc_semantic_nusc_tsd_full.py:474):
the batch index is appended as the LAST coordinate column."""
import numpy as np
import torch

from ..tensor import SparseTensor

__all__ = ['sparse_collate', 'sparse_collate_fn']


def sparse_collate(inputs):
    coords, feats = [], []
    stride = inputs[0].stride
    for k, x in enumerate(inputs):
        if isinstance(x.coords, np.ndarray):
            x.coords = torch.tensor(x.coords)
        if isinstance(x.feats, np.ndarray):
            x.feats = torch.tensor(x.feats)
        assert isinstance(x.coords, torch.Tensor), type(x.coords)
        assert isinstance(x.feats, torch.Tensor), type(x.feats)
        assert x.stride == stride, (x.stride, stride)
        input_size = x.coords.shape[0]
        batch = torch.full((input_size, 1), k, device=x.coords.device, dtype=torch.int)
        coords.append(torch.cat((x.coords, batch), dim=1))
        feats.append(x.feats)
    return SparseTensor(coords=torch.cat(coords, dim=0), feats=torch.cat(feats, dim=0), stride=stride)


def sparse_collate_fn(inputs):
    if isinstance(inputs[0], dict):
        output = {}
        for name in inputs[0].keys():
            if isinstance(inputs[0][name], dict):
                output[name] = sparse_collate_fn([x[name] for x in inputs])
            elif isinstance(inputs[0][name], np.ndarray):
                output[name] = torch.stack([torch.tensor(x[name]) for x in inputs], dim=0)
            elif isinstance(inputs[0][name], torch.Tensor):
                output[name] = torch.stack([x[name] for x in inputs], dim=0)
            elif isinstance(inputs[0][name], SparseTensor):
                output[name] = sparse_collate([x[name] for x in inputs])
            else:
                output[name] = [x[name] for x in inputs]
        return output
    return inputs
