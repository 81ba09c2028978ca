"""Does stream priority help the step?  The pipelined KD step issued from a HIGH-priority stream (the student's LiDAR chain), the
camera / teacher / weight-gradient / geometry streams at normal priority -- against the default (everything normal).
  python tools/exp_priority.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)


def measure(label, ctx):
    with ctx:
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(24):
            step()
        torch.cuda.synchronize()
    print('%-40s %.2f ms/step' % (label, (time.perf_counter() - t0) / 24 * 1e3), flush=True)


import contextlib
measure('default stream', contextlib.nullcontext())
hp = torch.cuda.Stream(priority=-1)
hp.wait_stream(torch.cuda.current_stream())
measure('step issued from a high-priority stream', torch.cuda.stream(hp))
torch.cuda.current_stream().wait_stream(hp)
measure('default stream again', contextlib.nullcontext())
