"""Host time per phase of the pipelined KD step (bench loop: fresh batches, geometry prefetch): forward issue (teacher / student /
camera head inside it), losses, backward + optimizer issue, the next batch's geometry (with its blocking size reads).
python tools/host_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse()      # (host_phases.py --voxels 3000 --image-hw 64 112: the host floor, a scene too small to load the GPU)
args.steps, args.warmup = 16, 6
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
run = step.runner
acc = {}


def timed(obj, name, label):
    real = getattr(obj, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return real(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, wrap)


timed(run.model, 'prepare', 'geometry of the next batch (incl. blocking reads)')
from u2mkd_amd.torchsparse.nn import functional as spf
import u2mkd_amd.train as T0
timed(spf, 'wait_counts', '  of which: waiting for posted sizes (wait_counts)')
_slice = [0]
_real_adv = T0._advance_geometry


def _adv(*a, **k):
    t0 = time.perf_counter()
    try:
        return _real_adv(*a, **k)
    finally:
        lab = 'geometry slice %d (launches + its read)' % (_slice[0] % 3 + 1)
        _slice[0] += 1
        acc[lab] = acc.get(lab, 0.0) + time.perf_counter() - t0


T0._advance_geometry = _adv
timed(run.model.model_t, 'forward', 'teacher forward issue')
timed(run.model.model_s, 'forward', 'student forward issue (incl. camera)')
timed(run.amp, 'backward_and_step', 'backward + optimizer issue')
from u2mkd_amd import kd as KD
real_losses = KD.kd_losses


def losses(*a, **k):
    t0 = time.perf_counter()
    try:
        return real_losses(*a, **k)
    finally:
        acc['losses issue'] = acc.get('losses issue', 0.0) + time.perf_counter() - t0
KD.kd_losses = losses
import u2mkd_amd.train as T
T.KD.kd_losses = losses
for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print('per step: host %.2f ms, wall %.2f ms' % (host / args.steps * 1e3, wall / args.steps * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-55s %6.2f ms' % (k, v / args.steps * 1e3))
print('  %-55s %6.2f ms' % ('everything else (batch copies, bookkeeping)', (host - sum(acc.values())) / args.steps * 1e3))
