"""Timing experiment: the tile kernel with and without its MFMAs (kflip bit 3)."""
import sys; sys.path.insert(0, '.')
import torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev
b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,)*3, (3,)*3, (1,)*3)
n = km.n_out
sch = km.schedule(False)
for cin, cout in ((64, 64), (32, 32)):
    x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda')
    wt = F._transpose_weights(w); o = torch.empty(n, cout, device='cuda'); st = L.stream()
    for flag in (0, 8):
        t = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(sch.nbr_s), L.ptr(sch.order), L.ptr(sch.tile_order), n, 27, flag, 0, L.ptr(o), st))
        print(f'{cin}->{cout} kflip={flag}: {t*1e3:.1f} us', flush=True)
