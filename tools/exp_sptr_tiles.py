"""Isolated timing of the spherical branch's window attention at the four strides of the 80 000-point scene (stride 2 .. 16:
the shapes of the KD step's SphereFormer blocks), forward per-pair kernels (csrc/sptr.hip) against the tile form
(csrc/sptr_tiles.hip), and the backward; nothing else runs on the GPU.
  python tools/exp_sptr_tiles.py [heads_at_stride_2=1]   (student: 1, 2, 4, 8 sphere heads; teacher: 2, 4, 8, 16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from u2mkd_amd import sptr
from u2mkd_amd.sptr import functional as SF
from u2mkd_amd.synth import synth_batch
from u2mkd_amd.lidar.spvcnn_spformer import spformer_kwargs
from u2mkd_amd.lidar.sphereformer import cart2sphere
from tools.ab_conv import ev

h0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
kw = spformer_kwargs()
d, qgl = 16, 24
wss = np.array(kw['window_size_sphere'], dtype=np.float64)
qss = np.array(kw['quant_size_sphere'], dtype=np.float64)
for stage in range(4):
    s = 2 << stage
    cs = torch.unique(torch.div(c[:, :3], s, rounding_mode='floor').int(), dim=0)
    xyz = (cs.float() + 0.5) * s * 0.05
    xyz = xyz - xyz.mean(0)
    n = xyz.shape[0]
    bi = torch.zeros(n, dtype=torch.int32, device='cuda')
    pts = cart2sphere(xyz).contiguous()
    window, quant = wss.copy(), qss.copy()
    window[:2] *= 2.0 ** stage
    quant[:2] *= 2.0 ** stage
    h = h0 << stage
    L = 2 * qgl
    plan = sptr.WindowPlan(pts, bi, window)
    wl = plan.wlen.float()
    q, k, v = (torch.randn(n, h, d, device='cuda', requires_grad=True) for _ in range(3))
    tq, tk, tv = ((0.3 * torch.randn(L, 3, h, d, device='cuda')).requires_grad_(True) for _ in range(3))
    go = torch.randn(n, h, d, device='cuda')
    res = {}
    for mode in ('0', 'all'):
        SF._TILES = mode
        f = lambda: sptr.window_attention(q, k, v, pts, plan, quant, qgl, tq, tk, tv, kw['a'])
        with torch.no_grad():
            res[mode] = ev(f, 10)
        if mode == '0':
            def fb():
                f().backward(go)
            tfb = ev(fb, 5)
    pairs = float((wl).sum())
    print('stride %2d: n=%6d h=%2d pairs=%.2fM window mean %.0f max %d | fwd per-pair %.0f us, tiles %.0f us | bwd %.0f us' % (
        s, n, h, pairs / 1e6, float(wl.mean()), int(wl.max()), res['0'] * 1e3, res['all'] * 1e3, (tfb - res['0']) * 1e3), flush=True)
