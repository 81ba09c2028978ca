"""Tile-local schedule (conv_tp) against the global pair schedule (conv_px3) on the borderline layer shapes, forward and
input gradient through functional._conv_os (U2MKD_CONV_SCHEDULE forces either).   python tools/ab_schedule.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ts_ref as R
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
from tools.ab_conv import ev

coords = synth_batch(80000, 1)['coords']; ts = 1
levels = {}
for lv in range(4):
    levels[ts] = coords
    coords = R.spdownsample(coords, 2, 2, ts); ts *= 2
shapes = [(1, 64, 128), (2, 64, 128), (4, 64, 128), (8, 64, 128), (1, 128, 64), (4, 128, 64), (8, 128, 64)] if len(sys.argv) > 1 else [(1, 64, 96), (1, 96, 64), (1, 96, 96), (1, 128, 96), (1, 96, 128), (2, 96, 128), (2, 128, 128), (2, 128, 64), (4, 128, 128)]
kms = {}
for ts, cin, cout in shapes:
    if ts not in kms:
        kms[ts] = F.build_kmap(torch.from_numpy(levels[ts]).cuda(), (ts,) * 3, (3,) * 3, (1,) * 3)
    km = kms[ts]; n = km.n_out
    x = torch.randn(n, cin, device='cuda'); g = torch.randn(n, cout, device='cuda')
    w = torch.randn(27, cin, cout, device='cuda') / (27 * cin) ** 0.5
    res = []
    for mode in ('tiles', 'pairs'):
        os.environ['U2MKD_CONV_SCHEDULE'] = mode
        try:
            tf = ev(lambda: F._conv_os(x, w, True, cout, km, False, n, 0))
            tb = ev(lambda: F._conv_os(g, w, False, cin, km, False, n, 1))
            res.append('%s fwd %.0f us dgrad %.0f us' % (mode, tf * 1e3, tb * 1e3))
        except Exception as e:
            res.append('%s: %s' % (mode, str(e)[:60]))
    print('ts=%d N=%d %d->%d: %s' % (ts, n, cin, cout, ' | '.join(res)), flush=True)
