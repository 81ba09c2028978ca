"""MIOpen on the camera branch's convolutions, NCHW against channels_last ("channels_last for the
MIOpen convs"): every Conv2d of SwiftNetRes18 at the KD step's size (6 x 3 x 360 x 640), forward + backward (data and
weight gradients), summed per layout.  The shapes are recorded from one forward of the real module."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from u2mkd_amd.camera import SwiftNetRes18

net = SwiftNetRes18(num_feature=(128, 128, 128)).cuda().train()
shapes = []
hooks = [m.register_forward_hook(lambda m, a, o: shapes.append((m, tuple(a[0].shape)))) for m in net.modules() if isinstance(m, nn.Conv2d)]
x = torch.randn(6, 3, 360, 640, device='cuda')
with torch.no_grad():
    net(x)
for h in hooks:
    h.remove()


def t(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = {False: [0.0, 0.0], True: [0.0, 0.0]}
for m, shp in shapes:
    row = []
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        xi = torch.randn(shp, device='cuda').contiguous(memory_format=fmt).requires_grad_(shp[1] != 3)
        w = m.weight.detach().clone().contiguous(memory_format=fmt).requires_grad_(True)
        y = F.conv2d(xi, w, None, m.stride, m.padding, m.dilation, m.groups)
        g = torch.randn_like(y)
        f = t(lambda: F.conv2d(xi, w, None, m.stride, m.padding, m.dilation, m.groups))
        ins = [w] + ([xi] if xi.requires_grad else [])
        b = t(lambda: torch.autograd.grad(y, ins, g, retain_graph=True))
        tot[cl][0] += f; tot[cl][1] += b
        row += [f, b]
    print('%-28s k%d s%d  NCHW fwd %.3f bwd %.3f | NHWC fwd %.3f bwd %.3f ms' % (str(shp), m.kernel_size[0], m.stride[0], *row), flush=True)
print('TOTAL NCHW fwd %.2f bwd %.2f = %.2f ms | channels_last fwd %.2f bwd %.2f = %.2f ms'
      % (tot[False][0], tot[False][1], sum(tot[False]), tot[True][0], tot[True][1], sum(tot[True])))
