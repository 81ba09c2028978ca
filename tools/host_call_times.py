"""Host time inside every C-ABI entry point (ctypes call gate included) per pipelined KD step: total, count, mean and the
slowest single call by entry -- a launch that takes 100x its usual few microseconds is a launch that WAITED (a full hardware
queue, the runtime's kernel-argument ring, a hidden synchronisation).  Also: the same for a few torch calls that can block.
  python tools/host_call_times.py [steps=16]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from u2mkd_amd import _lib as L

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
slow = []
real_call = L.call
pc = time.perf_counter
T0 = [0.0]


def call(name, *a):
    t = pc()
    real_call(name, *a)
    d = pc() - t
    e = acc[name]
    e[0] += d; e[1] += 1
    if d > e[2]:
        e[2] = d
    if d > 200e-6:
        slow.append((t - T0[0], d, name))


L.call = call


def wrap(obj, name, label):
    real = getattr(obj, name)

    def f(*a, **k):
        t = pc()
        try:
            return real(*a, **k)
        finally:
            d = pc() - t
            e = acc[label]
            e[0] += d; e[1] += 1
            if d > e[2]:
                e[2] = d
            if d > 200e-6:
                slow.append((t - T0[0], d, label))
    setattr(obj, name, f)


wrap(torch, 'empty', 'torch.empty')
wrap(torch, 'empty_like', 'torch.empty_like')
wrap(torch.Tensor, 'clone', 'Tensor.clone')
wrap(torch.Tensor, 'record_stream', 'Tensor.record_stream')
wrap(torch.nn.functional, 'conv2d', 'F.conv2d')
wrap(torch.cuda.Stream, 'wait_event', 'Stream.wait_event')
wrap(torch.cuda.Stream, 'wait_stream', 'Stream.wait_stream')
wrap(torch.cuda.Stream, 'record_event', 'Stream.record_event')
for _ in range(6):
    step()
torch.cuda.synchronize()
acc.clear(); slow.clear()
T0[0] = t0 = pc()
marks = []
for _ in range(STEPS):
    marks.append(pc() - t0)
    step()
host = pc() - t0
torch.cuda.synchronize()
wall = pc() - t0
print('per step: host %.2f ms, wall %.2f ms' % (host / STEPS * 1e3, wall / STEPS * 1e3))
tot = sum(v[0] for k, v in acc.items() if k.startswith('u2mkd_'))
print('C-ABI calls: %.2f ms per step in %d calls' % (tot / STEPS * 1e3, sum(v[1] for k, v in acc.items() if k.startswith('u2mkd_')) / STEPS))
print('%-44s %9s %7s %9s %9s' % ('entry', 'ms/step', 'n/step', 'mean us', 'max us'))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print('%-44s %9.3f %7.1f %9.1f %9.1f' % (k, v[0] / STEPS * 1e3, v[1] / STEPS, v[0] / v[1] * 1e6, v[2] * 1e6))
print('\ncalls above 200 us: %d per step, %.2f ms per step; by position in the step (ms since step begin):' % (len(slow) / STEPS, sum(d for _, d, _ in slow) / STEPS * 1e3))
import bisect
for t, d, name in slow[:80]:
    i = bisect.bisect_right(marks, t) - 1
    print('  step %2d +%6.2f ms  %7.1f us  %s' % (i, (t - marks[i]) * 1e3, d * 1e6, name))
