"""Microbench of the window attention fwd/bwd on the 80k-voxel synthetic scene, cubic and
spherical branch, with the SphereFormer configuration (window 0.3 m / (2deg,2deg,120m), qgl 24)."""
import os, sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import sptr
from u2mkd_amd.synth import synth_batch
from u2mkd_amd.lidar.spvcnn_spformer import spformer_kwargs
from tools.ab_conv import ev

def main():
    b = synth_batch(80000, 1)
    c = torch.from_numpy(b['coords']).cuda()
    xyz = c[:, :3].float() * 0.05
    bi = c[:, 3].int()
    kw = spformer_kwargs()
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n, d, qgl = xyz.shape[0], 16, 24
    for sphere in (False,):
        if sphere:
            x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
            r = torch.sqrt(x * x + y * y + z * z).clamp(min=1e-6)
            pts = torch.stack([torch.rad2deg(torch.atan2(y, x)) % 360, torch.rad2deg(torch.acos((z / r).clamp(-1, 1))), r], 1)
            window, quant, a = np.array(kw['window_size_sphere'], dtype=np.float64), np.array(kw['quant_size_sphere']), kw['a']
        else:
            pts, window, quant, a = xyz, np.array(kw['window_size']), np.array(kw['quant_size']), None
        L = 2 * qgl if sphere else 2 * qgl - 1
        plan = sptr.WindowPlan(pts, bi, window)
        wl = plan.wlen.float()
        q, k, v = (torch.randn(n, h, d, device='cuda', requires_grad=True) for _ in range(3))
        tq, tk, tv = (0.3 * torch.randn(L, 3, h, d, device='cuda')).requires_grad_(True), (0.3 * torch.randn(L, 3, h, d, device='cuda')).requires_grad_(True), (0.3 * torch.randn(L, 3, h, d, device='cuda')).requires_grad_(True)
        go = torch.randn(n, h, d, device='cuda')
        out = sptr.window_attention(q, k, v, pts, plan, quant, qgl, tq, tk, tv, a)
        tf = ev(lambda: sptr.window_attention(q, k, v, pts, plan, quant, qgl, tq, tk, tv, a), 10)
        def fb():
            o = sptr.window_attention(q, k, v, pts, plan, quant, qgl, tq, tk, tv, a)
            o.backward(go)
        tfb = ev(fb, 10)
        print(f'sphere={sphere} n={n} h={h} pairs={int(wl.sum())} wl mean {wl.mean():.1f} max {int(wl.max())} (pair-weighted mean {float((wl*wl).sum()/wl.sum()):.1f}): fwd {tf*1e3:.0f}us  fwd+bwd {tfb*1e3:.0f}us  -> bwd {1e3*(tfb-tf):.0f}us', flush=True)

main()
