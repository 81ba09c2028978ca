"""Window attention at the reference harness size (third_party/SparseTransformer/test/test_attention_op_step1.py:
8-15 and test_relative_pos_encoding_op_step2.py: N = 35 000 tokens in n = 1 500 windows, h = 6 heads of 16
channels): forward and forward + backward times, pairs M = sum of squared window lengths, algorithmic bytes per
SURVEY.md section 8d (q, k, v, out rows + coordinates / indices; the fused kernels never write M-sized arrays) and
what the reference's unfused dataflow would move for the same call (index_0/1 [M], rel_idx [M,3], attn [M,h] twice).
Run it under rocprofv3 --kernel-trace --stats for the per-kernel figures in profiles/."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from u2mkd_amd import sptr
from tools.ab_conv import ev

torch.manual_seed(1)
N, n_win, h, d, qgl = 35000, 1500, 6, 16, 24
win = torch.randint(0, n_win, (N,), device='cuda')
# one window per unit cell along x; positions inside the cell random (the quantised relative positions vary)
pts = torch.stack([win.float() + torch.rand(N, device='cuda') * 0.999, torch.rand(N, device='cuda') * 0.999,
                   torch.rand(N, device='cuda') * 0.999], 1)
batch = torch.zeros(N, dtype=torch.int32, device='cuda')
plan = sptr.WindowPlan(pts, batch, np.array([1.0, 1.0, 1.0]))
wl = plan.wlen.float()
M = int((wl).sum())                       # = sum over windows of L_w^2 (every token sees its whole window)
quant = np.array([1.0 / qgl] * 3)
L = 2 * qgl - 1
q, k, v = (torch.randn(N, h, d, device='cuda', requires_grad=True) for _ in range(3))
tq, tk, tv = ((0.3 * torch.randn(L, 3, h, d, device='cuda')).requires_grad_(True) for _ in range(3))
go = torch.randn(N, h, d, device='cuda')
fwd = lambda: sptr.window_attention(q, k, v, pts, plan, quant, qgl, tq, tk, tv, None)
def fb():
    fwd().backward(go)
tf, tfb = ev(fwd, 20), ev(fb, 20)
alg = N * h * d * 4 * 4 + N * (3 * 4 + 4 + 8)                       # section 8d: q, k, v, out + coords / window ranges
alg_bwd = N * h * d * 4 * 7 + N * (3 * 4 + 4 + 8) + 3 * L * 3 * h * d * 4 * 2   # + dout, dq, dk, dv; tables and their gradients
unfused = M * (2 * 4 + 3 * 4) + N * 3 * h * d * 4 + N * h * d * 4 + M * h * 4 * 2
print(f'N={N} windows={n_win} h={h} M={M} (mean window {float(wl.mean()):.1f}, max {int(wl.max())})')
print(f'fwd {tf*1e3:.1f} us: algorithmic {alg/1e6:.2f} MB -> {alg/(tf*1e-3)/1e9:.0f} GB/s = {alg/(tf*1e-3)/8e12:.4f} of 8 TB/s; '
      f'the unfused reference dataflow would move {unfused/1e6:.1f} MB')
print(f'fwd+bwd {tfb*1e3:.1f} us (bwd {1e3*(tfb-tf):.1f} us): algorithmic {(alg+alg_bwd)/1e6:.2f} MB -> {(alg+alg_bwd)/(tfb*1e-3)/1e9:.0f} GB/s')
print(f'pair rate: {M/(tf*1e-3)/1e9:.2f} G pairs/s forward ({h} heads: {M*h/(tf*1e-3)/1e9:.1f} G (pair, head)/s)')
