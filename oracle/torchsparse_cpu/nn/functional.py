"""torchsparse.nn.functional on CPU over oracle.ts_ref."""
import numpy as np
import torch
from torch.autograd import Function

from ... import ts_ref as R
from ..tensor import SparseTensor

__all__ = ['sphash', 'sphashquery', 'spcount', 'spvoxelize', 'spdevoxelize',
           'calc_ti_weights', 'spdownsample', 'conv3d']


def sphash(coords, offsets=None):
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    c = coords.contiguous().numpy()
    if offsets is None:
        return torch.from_numpy(R.sphash(c))
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    return torch.from_numpy(R.sphash(c, offsets.contiguous().numpy()))


def sphashquery(queries, references):
    return torch.from_numpy(R.sphashquery(queries.contiguous().numpy(),
                                          references.contiguous().numpy()))


def spcount(coords, num):
    return torch.from_numpy(R.spcount(coords.contiguous().numpy(), int(num)))


class _Voxelize(Function):
    @staticmethod
    def forward(ctx, feats, coords, counts):
        feats = feats.contiguous()
        coords = coords.contiguous().int()
        ctx.for_backwards = (coords, counts, feats.shape[0])
        return R.voxelize_forward(feats, coords, counts)

    @staticmethod
    def backward(ctx, grad_output):
        coords, counts, n = ctx.for_backwards
        return R.voxelize_backward(grad_output.contiguous(), coords, counts, n), None, None


def spvoxelize(feats, coords, counts):
    return _Voxelize.apply(feats, coords, counts)


class _Devoxelize(Function):
    @staticmethod
    def forward(ctx, feats, coords, weights):
        feats = feats.contiguous()
        coords = coords.contiguous().int()
        weights = weights.contiguous()
        ctx.for_backwards = (coords, weights, feats.shape[0])
        return R.devoxelize_forward(feats, coords, weights)

    @staticmethod
    def backward(ctx, grad_output):
        coords, weights, n = ctx.for_backwards
        return R.devoxelize_backward(grad_output.contiguous(), coords, weights, n), None, None


def spdevoxelize(feats, coords, weights):
    return _Devoxelize.apply(feats, coords, weights)


def calc_ti_weights(coords, idx_query, scale=1):
    with torch.no_grad():
        return R.calc_ti_weights(coords, idx_query, scale)


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    return torch.from_numpy(R.spdownsample(coords.numpy(), stride, kernel_size, tensor_stride))


class _Convolution(Function):
    @staticmethod
    def forward(ctx, input, weight, nbmaps, nbsizes, sizes, transposed=False):
        input = input.contiguous()
        weight = weight.contiguous()
        ctx.for_backwards = (input, weight, nbmaps, nbsizes, transposed)
        return R.conv_forward(input, weight, nbmaps, nbsizes, sizes, transposed)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, nbmaps, nbsizes, transposed = ctx.for_backwards
        gi, gw = R.conv_backward(input, weight, grad_output.contiguous(), nbmaps, nbsizes, transposed)
        return gi, gw, None, None, None, None


def conv3d(input, weight, kernel_size, bias=None, stride=1, dilation=1, transposed=False):
    feats, coords = input.feats, input.coords
    kernel_size = R.make_ntuple(kernel_size)
    stride = R.make_ntuple(stride)
    dilation = R.make_ntuple(dilation)

    if kernel_size == (1, 1, 1) and stride == (1, 1, 1) and dilation == (1, 1, 1):
        feats = feats.matmul(weight)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=coords, feats=feats, stride=input.stride)
    elif not transposed:
        key = (input.stride, kernel_size, stride, dilation)
        kmap = input.kmaps.get(key)
        if kmap is None:
            nbmaps, nbsizes, oc, _ = R.build_kmap(coords.numpy(), input.stride, kernel_size, stride)
            out_coords = coords if not any(s > 1 for s in stride) else torch.from_numpy(oc)
            kmap = [torch.from_numpy(nbmaps), torch.from_numpy(nbsizes),
                    (feats.shape[0], out_coords.shape[0]), out_coords]
            input.kmaps[key] = kmap
        out_coords = kmap[3]
        feats = _Convolution.apply(feats, weight, kmap[0], kmap[1], kmap[2], transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=out_coords, feats=feats,
                              stride=tuple(input.stride[k] * stride[k] for k in range(3)))
    else:
        tensor_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        kmap = input.kmaps[(tensor_stride, kernel_size, stride, dilation)]
        feats = _Convolution.apply(feats, weight, kmap[0], kmap[1], kmap[2], transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=input.cmaps[tensor_stride], feats=feats, stride=tensor_stride)

    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output
