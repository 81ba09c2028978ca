"""MIOpen at tensors beyond 2^31 bytes: the 7x7 stem convolution and the 1x1 pixel classifier of the KD student at
6 x 900 x 1600, batched against per-image evaluation (forward and weight gradient)."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
dev = 'cuda'
def check(name, x, w, stride, pad):
    x = x.to(dev); w = w.to(dev).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    g = torch.randn_like(y)
    y.backward(g)
    gw = w.grad.clone(); w.grad = None
    ys, gws = [], torch.zeros_like(w)
    for i in range(x.shape[0]):
        yi = F.conv2d(x[i:i + 1], w, None, stride, pad)
        yi.backward(g[i:i + 1])
        ys.append(yi.detach())
    gws = w.grad.clone()
    yc = torch.cat(ys)
    print('%-34s out %.2f GB | forward batched vs per image %.3g | weight gradient %.3g' % (
        name, y.numel() * 4 / 2 ** 30, float((y.detach() - yc).abs().max() / yc.abs().max()), float((gw - gws).norm() / gws.norm())), flush=True)
check('stem 7x7 3->64 @ 6x900x1600', torch.randn(6, 3, 900, 1600), torch.randn(64, 3, 7, 7) * 0.05, 1, 3)
check('stem 7x7 3->64 @ 6x360x640', torch.randn(6, 3, 360, 640), torch.randn(64, 3, 7, 7) * 0.05, 1, 3)
check('classifier 1x1 128->17 @ 6x900x1600', torch.randn(6, 128, 900, 1600), torch.randn(17, 128, 1, 1) * 0.05, 1, 0)
check('classifier 1x1 128->17 @ 3x900x1600', torch.randn(3, 128, 900, 1600), torch.randn(17, 128, 1, 1) * 0.05, 1, 0)
check('classifier 1x1 128->17 @ 6x360x640', torch.randn(6, 128, 360, 640), torch.randn(17, 128, 1, 1) * 0.05, 1, 0)
