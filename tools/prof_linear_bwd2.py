"""Where the host time of LinearFunction.backward / ConvolutionFunction.backward goes (they run on the autograd engine's thread,
out of cProfile's sight): wall-clock accumulators around their helpers.   python tools/prof_linear_bwd2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from u2mkd_amd import deferred, torchsparse as ts, _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
import u2mkd_amd.torchsparse.nn as spnn
from u2mkd_amd.synth import synth_batch

acc = {}


def wrap(obj, name, label=None):
    real = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return real(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
    return real


torch.manual_seed(0)
n = 20000
x = torch.randn(n, 64, device='cuda', requires_grad=True)
w = torch.nn.Parameter(torch.randn(96, 64, device='cuda') * 0.1)
b = torch.nn.Parameter(torch.zeros(96, device='cuda'))
coords = torch.from_numpy(np.ascontiguousarray(synth_batch(n, 1, seed=2)['coords'])).cuda()
conv = spnn.Conv3d(64, 64, 3).cuda()
xs = torch.randn(coords.shape[0], 64, device='cuda', requires_grad=True)
st = ts.SparseTensor(xs, coords)
conv(st)

wrap(F.LinearFunction, 'backward', 'LinearFunction.backward (whole)')
wrap(F.ConvolutionFunction, 'backward', 'ConvolutionFunction.backward (whole)')
wrap(F, '_wgrad_side')
wrap(F, '_dense_x3')
wrap(F, '_identity_pairs')
wrap(F, '_conv_os')
wrap(F, '_weight_layout')
wrap(L, 'call', 'L.call')
wrap(torch, 'empty', 'torch.empty')
wrap(torch, 'empty_like', 'torch.empty_like')
wrap(torch, 'sum', 'torch.sum')
wrap(deferred, 'side_for', 'deferred.side_for')
F.L.call = L.call


def lin():
    w.grad = None; b.grad = None
    F.linear(x, w, b).sum().backward()


def cv():
    conv.kernel.grad = None
    conv(st).F.sum().backward()


for name, fn in (('linear', lin), ('conv3d', cv)):
    with deferred.scope():
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
    print('=====', name, ': %.1f us of host time per forward + backward' % (host / 200 * 1e6))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print('   %-45s %7.1f us per iteration' % (k, v / 200 * 1e6))
