"""Which fp32 index arithmetic does torch's bilinear up-sampling (align_corners=True) follow on this build?  Candidates
evaluated in numpy against F.interpolate on one row."""
import numpy as np
import torch
import torch.nn.functional as F

for n_in, n_out in ((50, 100), (100, 200), (14, 28), (8, 16)):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n_in).astype(np.float32)
    y = F.interpolate(torch.from_numpy(x).cuda().view(1, 1, 1, n_in), (1, n_out), mode='bilinear', align_corners=True).cpu().numpy().ravel()
    ox = np.arange(n_out)
    x64 = x.astype(np.float64)

    def ev(src, lam=None):
        i0 = np.floor(src).astype(np.int64)
        i0 = np.minimum(i0, n_in - 1)
        ip = (i0 < n_in - 1).astype(np.int64)
        l1 = (src - i0) if lam is None else lam(i0)
        l1 = l1.astype(np.float32)
        l0 = (np.float32(1) - l1).astype(np.float32)
        return (l0.astype(np.float64) * x64[i0] + l1.astype(np.float64) * x64[i0 + ip])

    r32 = np.float32(n_in - 1) / np.float32(n_out - 1)
    cands = {
        'f32 r, f32 product': ev((r32 * ox.astype(np.float32)).astype(np.float32)),
        'f32 r, fma fraction': ev((r32 * ox.astype(np.float32)).astype(np.float32), lambda i0: np.float64(r32) * ox - i0),
        'f64 everything': ev((n_in - 1) / (n_out - 1) * ox),
        'f32 1/scale': ev((ox.astype(np.float32) / (np.float32(n_out - 1) / np.float32(n_in - 1))).astype(np.float32)),
        'half-pixel': ev(np.maximum((ox + 0.5) * n_in / n_out - 0.5, 0)),
    }
    for k, v in cands.items():
        print(n_in, n_out, '%-22s max abs diff %.3e' % (k, np.abs(v - y).max()))
