"""The wide layers' gather-sum with and without the BatchNorm statistics in its store, and the BatchNorm forward that follows
(statistics pass + merge + apply against merge + apply), in isolation, at the student's layer shapes of the 80 000-point scene.
  python tools/exp_gather_sum_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from u2mkd_amd import _lib as L
from u2mkd_amd.synth import synth_batch
from u2mkd_amd.torchsparse.nn import functional as F
from tools.ab_conv import ev

b = synth_batch(80000, 1)
c0 = torch.from_numpy(b['coords']).cuda()
lib = L.load()
rows = int(lib.u2mkd_pairs_gather_sum_stats_slab_rows())
for stride, cout in ((2, 64), (2, 128), (4, 128), (4, 256), (8, 256), (16, 256), (1, 96)):
    cs = torch.unique(torch.div(c0[:, :3], stride, rounding_mode='floor').int(), dim=0) * stride
    coords = torch.cat([cs, torch.zeros(len(cs), 1, dtype=torch.int32, device='cuda')], 1).contiguous()
    km = F.build_kmap(coords, (stride,) * 3, (3, 3, 3), (1, 1, 1))
    ps = km.pair_schedule()
    n = km.n_out
    y = torch.randn(ps.cap, cout, device='cuda')
    out = torch.empty(n, cout, device='cuda')
    st = L.stream()
    partial = torch.empty((n + rows - 1) // rows * 2 * cout, device='cuda')
    t_flat = ev(lambda: L.call('u2mkd_pairs_gather_sum', L.ptr(y), L.ptr(ps.pos_out), n, ps.k, cout, L.ptr(out), st), 50)
    t_stats = ev(lambda: L.call('u2mkd_pairs_gather_sum_stats', L.ptr(y), L.ptr(ps.pos_out), n, ps.k, cout, L.ptr(out), L.ptr(partial), st), 50)
    g = torch.ones(cout, device='cuda'); bt = torch.zeros(cout, device='cuda')
    rm = torch.zeros(cout, device='cuda'); rv = torch.ones(cout, device='cuda')
    mean = torch.empty(cout, device='cuda'); inv = torch.empty(cout, device='cuda'); yy = torch.empty_like(out)
    p2 = torch.empty(max(int(lib.u2mkd_bn_num_slabs(n)), 1) * 2 * cout, device='cuda')
    t_bn = ev(lambda: L.call('u2mkd_bn_train_forward_res', L.ptr(out), None, n, cout, L.ptr(g), L.ptr(bt), 1e-5, 0.1, L.ptr(rm), L.ptr(rv), None, 1,
                             L.ptr(p2), L.ptr(mean), L.ptr(inv), L.ptr(yy), st), 50)
    t_bnp = ev(lambda: L.call('u2mkd_bn_train_forward_from_partial', L.ptr(out), None, n, cout, L.ptr(g), L.ptr(bt), 1e-5, 0.1, L.ptr(rm), L.ptr(rv), None, 1,
                              L.ptr(partial), rows, L.ptr(mean), L.ptr(inv), L.ptr(yy), st), 50)
    print('stride %2d: n=%6d cout=%3d | gather-sum %5.1f us, with statistics %5.1f | BatchNorm forward %5.1f us, from partials %5.1f | sum %5.1f -> %5.1f'
          % (stride, n, cout, t_flat * 1e3, t_stats * 1e3, t_bn * 1e3, t_bnp * 1e3, (t_flat + t_bn) * 1e3, (t_stats + t_bnp) * 1e3), flush=True)
