"""Weight-gradient kernel on the GPU: the real pair list vs lists of the same shape whose rows are
sequential (no gather) -- separates gather latency from the MFMA / LDS pipeline."""
import sys; sys.path.insert(0, '.')
import torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev

b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,) * 3, (3,) * 3, (1,) * 3)
n = km.n_out
pairs, nbsizes, plan = km.pairs_plan()
P = int(plan[0])
lib = L.load(); st = L.stream()
for cin, cout in ((64, 64), (32, 32), (96, 96), (128, 128), (256, 256)):
    x = torch.randn(n, cin, device='cuda'); gy = torch.randn(n, cout, device='cuda')
    nb = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, 27)
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda'); dw = torch.empty(27, cin, cout, device='cuda')
    res = [f'{cin}->{cout} P={P} plan ch={int(plan[1])} wgs={int(plan[-1])}:']
    seq = torch.arange(pairs.shape[0], dtype=torch.int32, device='cuda') % n
    variants = {'real': pairs, 'seq-both': torch.stack([seq, seq], 1).contiguous(),
                'seq-in': torch.stack([seq, pairs[:, 1]], 1).contiguous()}
    for nm, pr in variants.items():
        t = ev(lambda: L.call('u2mkd_conv_wgrad_pairs', L.ptr(x), cin, L.ptr(gy), cout, L.ptr(pr), L.ptr(plan), n, 27, 0,
                              L.ptr(ws), nb, L.ptr(dw), st))
        res.append(f'{nm} {t*1e3:.1f}us ({2.0*P*cin*cout/(t*1e-3)/1e12:.1f}TF)')
    print(' | '.join(res), flush=True)
