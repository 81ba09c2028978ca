"""Stem MaxPool2d(3, 2, 1) at the KD step's size: torch's kernels against csrc/pixhead.hip, standalone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from u2mkd_amd import camera


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for shape in [(6, 64, 180, 320), (6, 64, 450, 800)]:
    x = torch.randn(shape, device='cuda').relu_().requires_grad_(True)
    mp = camera.MaxPool3x3s2()
    y = mp(x)
    g = torch.randn_like(y)
    yt = F.max_pool2d(x, 3, 2, 1)
    print(shape, 'hip fwd %.0f us bwd %.0f us | torch fwd %.0f us bwd %.0f us' % (
        t(lambda: mp(x)), t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)),
        t(lambda: F.max_pool2d(x, 3, 2, 1)), t(lambda: torch.autograd.grad(yt, x, g, retain_graph=True))))
