"""Time the three launches of the hybrid schedule separately (64->64 at 80k voxels)."""
import os, sys; sys.path.insert(0, '.')
import torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev

b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,)*3, (3,)*3, (1,)*3)
n = km.n_out
for cin, cout in ((64, 64), (96, 96)):
    x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda') / (27*cin)**0.5
    wt = F._transpose_weights(w); st = L.stream()
    for thr in (0, 16, 128, 1024, 10**9):
        os.environ['U2MKD_RARE_THR'] = str(thr)
        sch = F.ConvSchedule(km.nbr); out = torch.empty(n, cout, device='cuda')
        res = [f'{cin}->{cout} thr={thr} R={sch.n_rare} Pp={sch.p_pad}']
        if sch.n_rare:
            y = torch.empty(sch.p_pad, cout, device='cuda')
            t = ev(lambda: L.call('u2mkd_conv_forward_pairs', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(sch.pair_in), L.ptr(sch.tile_k), sch.p_pad, 27, 0, 432, L.ptr(y), st))
            res.append(f'pairs {t*1e3:.0f}us')
            t = ev(lambda: L.call('u2mkd_pairs_gather_sum', L.ptr(y), L.ptr(sch.pos), sch.n_rare, L.ptr(sch.order), sch.n_rare, 27, cout, L.ptr(out), st))
            res.append(f'gsum {t*1e3:.0f}us')
        if sch.n_rare < n:
            for var in (0, 432, 464):
                if var % 100 > cin: continue
                t = ev(lambda: L.call('u2mkd_conv_forward_rows', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(sch.nbr_s), n, L.ptr(sch.order), sch.n_rare, n, 27, 0, var, L.ptr(out), st))
                res.append(f'main v{var} {t*1e3:.0f}us')
        t = ev(lambda: sch.run(x, wt, cout, 0, out))
        res.append(f'all {t*1e3:.0f}us')
        print(' | '.join(res), flush=True)
