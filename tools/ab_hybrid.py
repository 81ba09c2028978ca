"""A/B the two conv schedules (output-stationary tiles vs offset-grouped pairs + gather-sum)
per SPVCNN layer shape on the GPU."""
import os, sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import ts_ref as R
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev


def main():
    b = synth_batch(80000, 1)
    coords = b['coords']; ts = 1
    levels = {}
    for lv in range(5):
        levels[ts] = coords
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2
    shapes = [(1, 64, 64), (1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 64, 64), (2, 64, 128), (2, 96, 96), (4, 128, 128), (8, 256, 256), (8, 384, 256), (16, 256, 256), (16, 512, 256)]
    pvars = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 32, 64]
    for (ts, cin, cout) in shapes:
        c = torch.from_numpy(levels[ts]).cuda()
        km = F.build_kmap(c, (ts,)*3, (3,)*3, (1,)*3)
        n = km.n_out; p = int((km.nbr >= 0).sum())
        x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda') / (27*cin)**0.5
        wt = F._transpose_weights(w); st = L.stream()
        sch, ps = km.schedule(False), km.pair_schedule()
        o1 = torch.empty(n, cout, device='cuda'); o2 = torch.empty_like(o1)
        t1 = ev(lambda: sch.run(x, wt, cout, 0, o1))
        res = [f'ts={ts} N={n} P={p} {cin}->{cout}: tiles {t1*1e3:.0f}us {2.0*p*cin*cout/(t1*1e-3)/1e12:.1f}TF']
        for var in pvars:
            if var == 64 and cin < 64: continue
            t2 = ev(lambda: ps.run(x, wt, cout, False, o2, variant=var))
            e = float((o2 - o1).abs().max())
            res.append(f'pairs v{var} {t2*1e3:.0f}us {2.0*p*cin*cout/(t2*1e-3)/1e12:.1f}TF e{e:.0e}')
        y = F._scratch(ps.cap * cout * 4, x.device)
        t3 = ev(lambda: L.call('u2mkd_pairs_gather_sum', L.ptr(y), L.ptr(ps.pos_out), n, 27, cout, L.ptr(o2), st))
        res.append(f'gather-sum {t3*1e3:.0f}us ({(p*cout*4+n*cout*4+n*27*4)/(t3*1e-3)/1e9:.0f} GB/s) auto={"pairs" if F._pairs_mode(cin, cout) else "tiles"}')
        print(' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
