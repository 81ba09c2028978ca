"""A/B on the GPU: wide-layer pair kernel in f32 MFMA (conv_pairs_kernel) vs bf16x3 (conv_px3_kernel), forward and
the swapped-role walk (input gradient), with the deviation between the two and against an fp64 reference of a
sample of rows."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ts_ref as R
from u2mkd_amd import _lib as L
L.LIB_PATH = L.LIB_PATH.replace('libu2mkd_hip.so', 'libu2mkd_hip%s.so' % os.environ.get('AB_SUFFIX', ''))
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
    shapes = shapes or [(1, 96, 96), (1, 128, 96), (2, 128, 128), (4, 192, 128), (4, 256, 256), (8, 256, 256), (8, 384, 256),
                        (8, 512, 512), (8, 768, 512), (16, 512, 512), (1, 192, 192)]
    kms = {}
    for (ts, cin, cout) in shapes:
        if ts not in kms:
            c = torch.from_numpy(levels[ts]).cuda()
            kms[ts] = F.build_kmap(c, (ts,) * 3, (3,) * 3, (1,) * 3)
        km = kms[ts]
        n = km.n_out; p = int((km.nbr >= 0).sum())
        x = torch.randn(n, cin, device='cuda'); w = torch.randn(27, cin, cout, device='cuda') / (27 * cin) ** 0.5
        ps = km.pair_schedule()
        res = [f'ts={ts} N={n} P={p} {cin}->{cout}:']
        for swap in (False, True):
            a = x if not swap else torch.randn(n, cout, device='cuda')
            ca, cb = (cin, cout) if not swap else (cout, cin)
            transpose = not swap
            wt = F._weight_layout(w, transpose, False)
            wf = F._weight_layout(w, transpose, True)
            o1 = torch.empty(n, cb, device='cuda'); o2 = torch.empty(n, cb, device='cuda')
            t1 = ev(lambda: ps.run(a, wt, cb, swap, o1))
            t2 = ev(lambda: ps.run(a, wf, cb, swap, o2, fragments=True))
            fl = 2.0 * p * cin * cout
            o3 = torch.empty(n, cb, device='cuda')
            ps.run(a, wf, cb, swap, o3, fragments=True)
            res.append(f"{'dgrad' if swap else 'fwd'} f32 {t1*1e3:.0f}us ({fl/(t1*1e-3)/1e12:.0f}TF) x3 {t2*1e3:.0f}us ({fl/(t2*1e-3)/1e12:.0f}TF) "
                       f"maxdiff {float((o1-o2).abs().max()):.1e} rel {float((o1-o2).abs().max()/o1.abs().max()):.1e} repro {bool(torch.equal(o2,o3))}")
        print(' | '.join(res), flush=True)


if __name__ == '__main__':
    # optional shapes on the command line: stride,cin,cout ...
    main([tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or None)
