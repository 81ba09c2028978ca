"""torch.profiler (CPU side) over a few pipelined KD steps: where the host's time goes by operator, own time.
python tools/host_torch_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(6):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        step()
torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in ev)
print('self CPU time per step (all threads): %.1f ms' % (tot / N / 1e3))
for e in rows[:45]:
    print('  %8.2f ms  %6.1f calls  %s' % (e.self_cpu_time_total / N / 1e3, e.count / N, e.key[:90]))
