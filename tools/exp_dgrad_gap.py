"""Why the roofline leg's dgrad launch is slower than its forward launch (same kernel, same table): time the tile kernel over
the combinations of input tensor, fragment orientation and offset flip.   python tools/exp_dgrad_gap.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import _lib as L
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch
import bench

coords = torch.from_numpy(synth_batch(80000, 1, seed=1234)['coords']).cuda()
km = F.build_kmap(coords, (1,) * 3, (3,) * 3, (1,) * 3)
n, cin, cout = km.n_out, 64, 64
sch = km.schedule(False)
lib = L.load(); st = L.stream()
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(n, cin, device='cuda', generator=g); gy = torch.randn(n, cout, device='cuda', generator=g)
w = torch.randn(27, cin, cout, device='cuda', generator=g) / (27 * cin) ** 0.5
wf = torch.empty(2, lib.u2mkd_weight_fragments_bytes(27, cin, cout, 0), dtype=torch.uint8, device='cuda')
L.call('u2mkd_weight_fragments', L.ptr(w), 27, cin, cout, 2, 0, L.ptr(wf), st)
o1 = torch.empty(n, cout, device='cuda'); o2 = torch.empty(n, cout, device='cuda')


def conv(a, frag, flip, o):
    L.call('u2mkd_conv_forward_tiles', L.ptr(a), n, cin, L.ptr(wf[frag]), cout, L.ptr(sch.nbr_s), L.ptr(sch.order),
           L.ptr(sch.items), L.ptr(sch.n_items), n, 27, flip, 0, L.ptr(o), st)


for rnd in range(2):
    for name, a, frag, flip, o in (('x  frag0 flip0 -> o1', x, 0, 0, o1), ('gy frag1 flip1 -> o2', gy, 1, 1, o2), ('x  frag0 flip1 -> o1', x, 0, 1, o1),
                                   ('x  frag1 flip0 -> o1', x, 1, 0, o1), ('gy frag0 flip0 -> o2', gy, 0, 0, o2), ('x  frag0 flip0 -> o2', x, 0, 0, o2),
                                   ('gy frag1 flip1 -> o1', gy, 1, 1, o1)):
        t = bench.time_events([lambda: conv(a, frag, flip, o)], 60)
        print('%s  %.1f us' % (name, t * 1e3), flush=True)
