"""What a millisecond is worth, by where it is spent: the pipelined KD step (bench loop: fresh batches, geometry prefetch) with D ms
of HOST busy-wait or of GPU spin (one workgroup on the main stream: chain latency, no throughput) injected at one point of the
step -- in front of the forward, in front of the backward, behind the step.  One process, the variants in turn on one model.
  python tools/exp_sensitivity.py [D=5] [steps=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

D = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sys.argv = sys.argv[:1]
args = bench.parse()
args.steps, args.warmup = STEPS, 6
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
run = step.runner
mode = {'host': None, 'gpu': None}


def spin_host(ms):
    t = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t:
        pass


# calibrate torch.cuda._sleep
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
CYC_PER_MS = 20_000_000 / e0.elapsed_time(e1)


def inject(where):
    if mode['host'] == where:
        spin_host(D)
    if mode['gpu'] == where:
        torch.cuda._sleep(int(D * CYC_PER_MS))


real_bwd = run.amp.backward_and_step


def bwd(loss, opt):
    inject('bwd')
    return real_bwd(loss, opt)


run.amp.backward_and_step = bwd


def one():
    inject('fwd')
    step()
    inject('end')


def measure(label):
    for _ in range(4):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        one()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('%-28s wall %.2f ms/step   host %.2f' % (label, wall / STEPS * 1e3, host / STEPS * 1e3), flush=True)


for _ in range(args.warmup):
    step()
measure('baseline')
for kind in ('host', 'gpu'):
    for where in ('fwd', 'bwd', 'end'):
        mode['host'] = mode['gpu'] = None
        mode[kind] = where
        measure('%s +%.0f ms @ %s' % (kind, D, where))
mode['host'] = mode['gpu'] = None
measure('baseline again')
