"""Time of the 64->64 stride-1 conv kernel vs number of kernel offsets visited (prefix of the
neighbour table): intercept = per-tile fixed cost, slope = per-offset cost."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev
b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,)*3, (3,)*3, (1,)*3)
n = km.n_out; st = L.stream()
nbr_s, order = km.sorted_table(False)
cin = cout = 64
x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda'); wt = F._transpose_weights(w)
o = torch.empty(n, cout, device='cuda')
valid = (nbr_s >= 0)
for var in (464, 3064):
    res = []
    for k in (1, 2, 4, 8, 13, 14, 20, 27):
        t = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order), None, n, k, 0, var, L.ptr(o), st), 20)
        pairs = int(valid[:k].sum())
        # active (64-row tile, offset) stages
        nt = (n + 63) // 64
        pad = nt * 64 - n
        act = torch.nn.functional.pad(valid[:k], (0, pad)).view(k, nt, 64).any(2).sum().item()
        res.append(f'k={k}: {t*1e3:.0f}us pairs={pairs} stages={act}')
    print(var, ' | '.join(res), flush=True)
