"""Host time per call of the hottest operators (forward, with autograd recording; backward through the engine), on inputs small
enough that the GPU never is the limit: what one application of the Python Function costs against the C++ one
(U2MKD_HOST_OPS=0 / 1; run once per setting).   python tools/exp_op_host_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import torchsparse as ts
from u2mkd_amd.torchsparse import nn as spnn
from u2mkd_amd.torchsparse.nn import functional as spf
from u2mkd_amd.lidar.blocks import PointLinear
from u2mkd_amd.synth import synth_batch

N = 2000
b = synth_batch(4000, 1, seed=3)
coords = torch.from_numpy(b['coords']).cuda()
x = torch.randn(coords.shape[0], 64, device='cuda', requires_grad=True)
bn = spnn.BatchNorm(64).cuda().train()
lin = PointLinear(64, 64).cuda()
conv = spnn.Conv3d(64, 64, 3).cuda()
st = ts.SparseTensor(x, coords)
conv(st)          # (builds the kernel map once)


def timed(label, fn, n=N):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%-44s %7.1f us per call (host)' % (label, t / n * 1e6), flush=True)


print('U2MKD_HOST_OPS =', os.environ.get('U2MKD_HOST_OPS', '1'))
timed('batch_norm + relu, forward (training)', lambda: spf.batch_norm(x, bn, True))
with torch.no_grad():
    timed('batch_norm + relu, forward (no grad)', lambda: spf.batch_norm(x, bn, True))
timed('linear 64->64, forward', lambda: lin(x))
timed('conv3d 64->64 k3, forward', lambda: conv(st))
g = torch.randn_like(x)


def fb(f):
    def run():
        y = f()
        y = y.F if hasattr(y, 'F') else y
        y.backward(g)
    return run


timed('batch_norm fwd + bwd', fb(lambda: spf.batch_norm(x, bn, True)), 500)
timed('linear fwd + bwd', fb(lambda: lin(x)), 500)
timed('conv3d fwd + bwd', fb(lambda: conv(st)), 500)
