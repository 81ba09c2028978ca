"""cProfile of nn.Linear / Conv3d forward + backward on the HIP path, host side only (where the 75 us per backward call go).
python tools/prof_linear_bwd.py"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from u2mkd_amd import deferred, torchsparse as ts
from u2mkd_amd.torchsparse.nn import functional as F
import u2mkd_amd.torchsparse.nn as spnn
from u2mkd_amd.synth import synth_batch

torch.manual_seed(0)
n = 20000
x = torch.randn(n, 64, device='cuda', requires_grad=True)
w = torch.nn.Parameter(torch.randn(96, 64, device='cuda') * 0.1)
b = torch.nn.Parameter(torch.zeros(96, device='cuda'))
coords = torch.from_numpy(np.ascontiguousarray(synth_batch(n, 1, seed=2)['coords'])).cuda()
conv = spnn.Conv3d(64, 64, 3).cuda()
xs = torch.randn(coords.shape[0], 64, device='cuda', requires_grad=True)
st = ts.SparseTensor(xs, coords)
conv(st)      # kernel map cached on the tensor


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
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            fn()
        pr.disable()
        torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
    print('=====', name, '(200 forward + backward)')
    print(s.getvalue()[:4200])
