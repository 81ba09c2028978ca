"""The price of a LAUNCH, wherever it runs: N one-element kernels (torch add_ on a 1-element tensor, ~2 us of GPU time each)
queued per KD step behind the step's call -- on the main stream (in line), on a stream of their own, or on one of the step's side
streams -- against the plain step.  N x 2 us of arithmetic nobody would notice; what the step pays is the dispatches.
  python tools/exp_empty_kernels.py [N=1000] [steps=16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from u2mkd_amd import kd as KD

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(8):
    step()
x = torch.zeros(1, device='cuda')
own = torch.cuda.Stream()
places = {'none': None, 'main': torch.cuda.current_stream(), 'own stream': own,
          'geo': KD._side_stream(x, 'geo'), 'teacher': KD._side_stream(x, 'teacher'), 'sparse_wgrad': KD._side_stream(x, 'sparse_wgrad')}
bufs = {k: torch.zeros(1, device='cuda') for k in places}


def measure(label, st):
    def one():
        step()
        if st is not None:
            with torch.cuda.stream(st):
                b = bufs[label]
                for _ in range(N):
                    b.add_(1.0)
    for _ in range(4):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        one()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('%-14s wall %.2f ms/step   host %.2f' % (label, wall / STEPS * 1e3, host / STEPS * 1e3), flush=True)


for label in ('none', 'main', 'none', 'own stream', 'none', 'geo', 'none', 'teacher', 'none', 'sparse_wgrad', 'none'):
    measure(label, places[label])
