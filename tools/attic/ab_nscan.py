"""Conv 64->64 k3 forward time vs scene size (latency-bound kernels do not scale with N)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev
st = L.stream(); cin = cout = 64
for n_vox in (10000, 20000, 40000, 80000, 160000, 320000):
    b = synth_batch(n_vox, 1)
    km = F.build_kmap(torch.from_numpy(b['coords']).cuda(), (1,)*3, (3,)*3, (1,)*3)
    n = km.n_out; p = int((km.nbr >= 0).sum())
    nbr_s, order = km.sorted_table(False)
    x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda'); wt = F._transpose_weights(w)
    o = torch.empty(n, cout, device='cuda')
    t = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order), None, n, 27, 0, 464, L.ptr(o), st), 20)
    print(f'N={n} P={p} kbar={p/n:.2f}: {t*1e3:.1f} us  ({2.0*p*cin*cout/(t*1e-3)/1e12:.1f} TF, {t*1e3/ (n/64/256):.1f} us per tile-round)', flush=True)
