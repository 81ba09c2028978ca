"""MFMA ceiling of the conv kernels' loop structure: a fully dense grid (all 27 offsets active,
no skip waste), so useful FLOPs == executed FLOPs."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from tools.ab_conv import ev
D = 36
g = np.stack(np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing='ij'), -1).reshape(-1, 3)
c = torch.from_numpy(np.concatenate([g, np.zeros((len(g), 1))], 1).astype(np.int32)).cuda()
km = F.build_kmap(c, (1,)*3, (3,)*3, (1,)*3)
n = km.n_out; p = int((km.nbr >= 0).sum()); st = L.stream()
for cin, cout in ((64, 64), (128, 128), (256, 256)):
    x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda'); wt = F._transpose_weights(w)
    o = torch.empty(n, cout, device='cuda')
    res = [f'dense N={n} P={p} {cin}->{cout}:']
    for var in (432, 464, 832, 864, 3032, 3064):
        t = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(km.nbr), None, None, n, 27, 0, var, L.ptr(o), st), 10)
        res.append(f'{var}: {t*1e3:.0f}us {2.0*p*cin*cout/(t*1e-3)/1e12:.1f}TF')
    print(' | '.join(res), flush=True)
