"""A/B the conv kernels on the GPU: v1 (natural order, weights from L1/L2) vs
v2 (LDS-staged weights) with identity and mask-sorted row order."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import ts_ref as R
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch

def ev(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters

def main():
    b = synth_batch(80000, 1)
    coords = b['coords']; ts = 1
    levels = {}
    for lv in range(5):
        levels[ts] = coords
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2
    st = L.stream()
    for (ts, cin, cout) in [(1, 64, 64), (1, 96, 96), (1, 32, 32), (2, 96, 96), (4, 128, 128), (8, 256, 256), (8, 384, 256), (16, 256, 256)]:
        c = torch.from_numpy(levels[ts]).cuda()
        km = F.build_kmap(c, (ts,)*3, (3,)*3, (1,)*3)
        n = km.n_out; p = int((km.nbr >= 0).sum())
        x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda') / (27*cin)**0.5
        wt = F._transpose_weights(w)
        o1 = torch.empty(n, cout, device='cuda'); o2 = torch.empty_like(o1); o3 = torch.empty_like(o1)
        nbr_s, order = km.sorted_table(False)
        tord = L.ptr(km.schedule(False).tile_order)
        t1 = ev(lambda: L.call('u2mkd_conv_forward', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(km.nbr), n, 27, 0, L.ptr(o1), st))
        t1s = ev(lambda: L.call('u2mkd_conv_forward', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(nbr_s), n, 27, 0, L.ptr(o2), st))
        res = [f'ts={ts} N={n} P={p} {cin}->{cout}: v1 {t1*1e3:.0f}us v1-sorted {t1s*1e3:.0f}us']
        for var in (0,):
            if var % 100 == 64 and cin < 64: continue
            if 3000 <= var < 5000 and cout % 64: continue
            if var >= 50000 and (var - 50000) // 1000 * 16 > cout: continue
            t3 = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order), tord, n, 27, 0, var, L.ptr(o3), st))
            e3 = float((o3 - o1).abs().max())
            tf = 2.0 * p * cin * cout / (t3 * 1e-3) / 1e12
            res.append(f'{var}: {t3*1e3:.0f}us {tf:.1f}TF e{e3:.0e}')
        if False:
            var = 3064
            for flag, nm in ((2, 'sameB'), (4, 'localA'), (6, 'both')):
                tt = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(nbr_s), L.ptr(order), tord, n, 27, flag, var, L.ptr(o3), st))
                res.append(f'{nm} {tt*1e3:.0f}us')
        t2 = ev(lambda: L.call('u2mkd_conv_forward_sorted', L.ptr(x), n, cin, L.ptr(wt), cout, L.ptr(km.nbr), None, None, n, 27, 0, 0, L.ptr(o2), st))
        res.append(f'ident-heur {t2*1e3:.0f}us')
        # wgrad: table-scan (v1) vs pair-list (v2)
        gy = torch.randn(n, cout, device='cuda')
        lib = L.load()
        nb1 = lib.u2mkd_conv_wgrad_workspace_bytes(n, cin, cout, 27); ws1 = torch.empty(nb1, dtype=torch.uint8, device='cuda')
        dw1 = torch.empty_like(w); dw2 = torch.empty_like(w)
        tw1 = ev(lambda: L.call('u2mkd_conv_wgrad', L.ptr(x), cin, L.ptr(gy), cout, L.ptr(km.nbr), n, 27, 1, 1, L.ptr(ws1), nb1, L.ptr(dw1), st))
        pairs, _, plan = km.pairs_plan()
        nb2 = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, 27); ws2 = torch.empty(nb2, dtype=torch.uint8, device='cuda')
        tw2 = ev(lambda: L.call('u2mkd_conv_wgrad_pairs', L.ptr(x), cin, L.ptr(gy), cout, L.ptr(pairs), L.ptr(plan), n, 27, 0, L.ptr(ws2), nb2, L.ptr(dw2), st))
        ew = float((dw2 - dw1).abs().max() / dw1.abs().max())
        res.append(f'wgrad v1 {tw1*1e3:.0f}us v2 {tw2*1e3:.0f}us ({2.0*p*cin*cout/(tw2*1e-3)/1e12:.1f}TF) relerr {ew:.1e} plan {plan[:2].tolist()} wgs {int(plan[-1])}')
        print(' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
