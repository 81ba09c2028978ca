"""Per-phase cycles of workgroup 0 of the 64x64 weight-gradient kernel (in-kernel s_memtime stamps)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,) * 3, (3,) * 3, (1,) * 3)
n = km.n_out
pairs, nbsizes, plan = km.pairs_plan()
lib = L.load()
x = torch.randn(n, 64, device='cuda'); gy = torch.randn(n, 64, device='cuda')
nb = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, 64, 64, 27)
ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
st = torch.zeros(256, dtype=torch.int64, device='cuda')
for _ in range(3):
    L.call('u2mkd_debug_wgrad_stamps', L.ptr(x), L.ptr(gy), L.ptr(pairs), L.ptr(plan), n, 27, L.ptr(ws), L.ptr(st), L.stream())
torch.cuda.synchronize()
s = st.cpu().numpy()
s = s[s > 0]
d = np.diff(s)
k = len(d) // 4 * 4
ph = d[:k].reshape(-1, 4)      # [issue loads, multiply, store(wait rows), barrier + odd chunk ...]
print('plan', plan[:2].tolist(), 'stamps', len(s))
print('per even chunk: issue-loads, multiply, store, (barrier + whole odd chunk + loop)')
print(np.median(ph, axis=0), '\n', ph[:6])
