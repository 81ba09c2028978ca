"""Weight gradient on layer shapes of the KD step (stride, cin, cout): time per launch (gradient kernel + slab reduce) and
the deviation from an fp64 product over a sample of offsets.   python tools/ab_wgrad_layers.py [stride,cin,cout ...]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ts_ref as R
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev


def main(shapes=None):
    coords = synth_batch(80000, 1)['coords']; ts = 1
    levels = {}
    for lv in range(5):
        levels[ts] = coords
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2
    shapes = shapes or [(1, 64, 64), (2, 128, 128), (4, 256, 256), (8, 256, 256), (8, 512, 512), (8, 768, 512), (16, 512, 512), (1, 192, 192), (4, 128, 256)]
    lib = L.load(); st = L.stream()
    kms = {}
    for (ts, cin, cout) in shapes:
        if ts not in kms:
            kms[ts] = F.build_kmap(torch.from_numpy(levels[ts]).cuda(), (ts,) * 3, (3,) * 3, (1,) * 3)
        km = kms[ts]; n = km.n_out
        pairs, nbsizes, plan = km.pairs_plan()
        P = int(plan[0])
        x = torch.randn(n, cin, device='cuda'); gy = torch.randn(n, cout, device='cuda')
        nb = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, 27)
        ws = torch.empty(nb, dtype=torch.uint8, device='cuda'); dw = torch.empty(27, cin, cout, device='cuda')
        run = lambda: L.call('u2mkd_conv_wgrad_pairs', L.ptr(x), cin, L.ptr(gy), cout, L.ptr(pairs), L.ptr(plan), n, 27, 0,
                             L.ptr(ws), nb, L.ptr(dw), st)
        t = ev(run)
        sizes = nbsizes.cpu().tolist(); off = [0]
        for v in sizes: off.append(off[-1] + v)
        err = 0.0
        for k in (0, 13, 26):
            pr = pairs[off[k]:off[k + 1]].long()
            ref = x[pr[:, 0]].double().t() @ gy[pr[:, 1]].double()
            err = max(err, float((dw[k].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
        d2 = dw.clone(); run(); torch.cuda.synchronize()
        print(f'ts={ts} N={n} P={P} {cin}x{cout}: {t*1e3:.0f} us ({2.0*P*cin*cout/(t*1e-3)/1e12:.0f} TF)  rel err vs fp64 {err:.1e}  repro {bool(torch.equal(d2, dw))}', flush=True)


if __name__ == '__main__':
    main([tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or None)
