"""A/B on the GPU: tile-local pair schedule (conv_tp, variant 0) vs the offset-walking tile kernels
(variant 1) vs the global pair schedule, forward and input gradient, over the SPVCNN layer shapes."""
import sys; sys.path.insert(0, '.')
import torch
from oracle import ts_ref as R
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev


def main(shapes=None):
    b = synth_batch(80000, 1)
    coords = b['coords']; ts = 1
    levels = {}
    for lv in range(5):
        levels[ts] = coords
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2
    st = L.stream()
    shapes = shapes or [(1, 64, 64), (1, 32, 32), (1, 64, 128), (2, 32, 32), (4, 64, 64), (4, 32, 64), (8, 64, 128)]
    kms = {}
    for (ts, cin, cout) in shapes:
        if ts not in kms:
            c = torch.from_numpy(levels[ts]).cuda()
            kms[ts] = F.build_kmap(c, (ts,) * 3, (3,) * 3, (1,) * 3)
        km = kms[ts]
        n = km.n_out; p = int((km.nbr >= 0).sum())
        x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda') / (27 * cin) ** 0.5
        wt = F._transpose_weights(w)
        sch = km.schedule(False)
        outs = {}
        res = [f'ts={ts} N={n} P={p} {cin}->{cout}:']
        lib = L.load()
        frag = {}
        for ar in (1, 2):
            for tr in (0, 1):
                buf = torch.empty(lib.u2mkd_weight_fragments_bytes(27, cin, cout, ar), dtype=torch.uint8, device='cuda')
                L.call('u2mkd_weight_fragments', L.ptr(w), 27, cin, cout, tr, ar, L.ptr(buf), st)
                frag[(ar, tr)] = buf
        def run(nm, a, ca, wts, cb, flip, o):
            if nm in ('tp', 'tpf32'):
                ar = 2 if nm == 'tp' else 1
                L.call('u2mkd_conv_forward_tiles', L.ptr(a), n, ca, L.ptr(frag[(ar, 1 - flip)]), cb, L.ptr(sch.nbr_s), L.ptr(sch.order),
                       L.ptr(sch.items), L.ptr(sch.n_items), n, 27, flip, ar, L.ptr(o), st)
            else:
                L.call('u2mkd_conv_forward_sorted', L.ptr(a), n, ca, L.ptr(wts[1]), cb, L.ptr(sch.nbr_s), L.ptr(sch.order),
                       L.ptr(sch.tile_order), n, 27, flip, L.ptr(o), st)
        for nm in ('tp', 'tpf32', 'walk'):
            o = torch.empty(n, cout, device='cuda')
            t = ev(lambda: run(nm, x, cin, (None, wt), cout, 0, o))
            outs[nm] = o
            res.append(f'{nm} {t*1e3:.1f}us ({2.0*p*cin*cout/(t*1e-3)/1e12:.1f}TF)')
        ps = km.pair_schedule()
        o = torch.empty(n, cout, device='cuda')
        t = ev(lambda: ps.run(x, wt, cout, False, o))
        outs['pairs'] = o
        res.append(f'pairs {t*1e3:.1f}us')
        res.append('err tp-walk %.1e tpf32-walk %.1e tp-pairs %.1e' % (float((outs['tp'] - outs['walk']).abs().max()), float((outs['tpf32'] - outs['walk']).abs().max()),
                                                        float((outs['tp'] - outs['pairs']).abs().max())))
        # input gradient (kflip = 1 on the same table, B = kernel as [cin][cout])
        g = torch.randn(n, cout, device='cuda')
        d = {}
        for nm in ('tp', 'walk'):
            o = torch.empty(n, cin, device='cuda')
            t = ev(lambda: run(nm, g, cout, (None, w), cin, 1, o))
            d[nm] = o
            res.append(f'dgrad-{nm} {t*1e3:.1f}us')
        res.append('err %.1e' % float((d['tp'] - d['walk']).abs().max()))
        a = outs['tp'].clone()
        run('tp', x, cin, (None, wt), cout, 0, outs['tp'])
        res.append('bitwise-repro %s' % bool(torch.equal(a, outs['tp'])))
        print(' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
