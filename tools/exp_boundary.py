"""What does the MAIN stream wait for at the boundary between two pipelined KD steps?  GPU time stamps (HIP events) of
  A  the main stream's position when step k's call returns (backward, optimizer: everything of step k queued),
  G  the geometry stream's position at the same moment (the end of batch k + 1's prepared geometry / plans),
  W  the weight-gradient stream's position at the same moment,
  P  the main stream in front of the student's first kernel of step k + 1 (behind the wait for the geometry, the teacher's fork, ...),
  S  behind the student's stem,
with the host's clock at each.  P - max(A, G, W) is what the main stream waits for the HOST; max(G, W) - A what it waits for the side streams.
  python tools/exp_boundary.py [steps=12]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from u2mkd_amd import kd as KD

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
run = step.runner
ms = run.model.model_s
for _ in range(8):
    step()
rows = []
cur = {}


def ev(stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    return e, time.perf_counter()


def pre(mod, inp):
    cur['P'] = ev()


def post(mod, inp, out):
    cur['S'] = ev()


h1 = ms.stem.register_forward_pre_hook(pre)
h2 = ms.stem.register_forward_hook(post)
x = torch.zeros(1, device='cuda')
geo = KD._side_stream(x, 'geo')
wg = KD._side_stream(x, 'sparse_wgrad')
base = ev()
prev = None
for i in range(STEPS):
    cur = {}
    cur['B'] = ev()
    step()
    cur['A'] = ev()
    cur['G'] = ev(geo)
    cur['W'] = ev(wg)
    rows.append(cur)
torch.cuda.synchronize()
h1.remove(); h2.remove()
t0e, t0h = base
print('step |  begin(main)   P(before stem)  S(after stem) | A(main end)   G(geo end)   W(wgrad end) | next P - max(A,G,W)   [GPU ms since start | host ms]')
for i, r in enumerate(rows):
    g = {k: t0e.elapsed_time(v[0]) for k, v in r.items()}
    h = {k: (v[1] - t0h) * 1e3 for k, v in r.items()}
    nxt = rows[i + 1] if i + 1 < len(rows) else None
    extra = ''
    if nxt is not None:
        pn = t0e.elapsed_time(nxt['P'][0])
        extra = 'P+1 = %.2f: %.2f after A, %.2f after G, %.2f after W; host at P+1: %.2f' % (
            pn, pn - g['A'], pn - g['G'], pn - g['W'], (nxt['P'][1] - t0h) * 1e3)
    print('%3d | B %.2f/%.2f  P %.2f/%.2f  S %.2f/%.2f | A %.2f/%.2f  G %.2f  W %.2f | %s' % (
        i, g['B'], h['B'], g['P'], h['P'], g['S'], h['S'], g['A'], h['A'], g['G'], g['W'], extra))
