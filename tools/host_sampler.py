"""Where does the host thread SIT during the pipelined KD step?  A sampler thread reads the main thread's Python stack every
~0.1 ms (it gets the GIL whenever the main thread is inside a call that releases it -- a blocked launch, a synchronising copy --
and at every switch interval otherwise) and counts the innermost frames of this repository and the innermost frame overall.
Blocking calls are over-represented against pure Python time by construction: this is a detector of WAITING, not a profiler.
  python tools/host_sampler.py [steps=24]"""
import collections
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(6):
    step()
torch.cuda.synchronize()

main_id = threading.main_thread().ident
inner, ours, phases = collections.Counter(), collections.Counter(), collections.Counter()
stop = [False]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gaps = []


def sampler():
    me = threading.get_ident()
    last = time.perf_counter()
    while not stop[0]:
        frames = sys._current_frames()
        now = time.perf_counter()
        gaps.append(now - last)
        last = now
        # the thread that is doing the step's work right now: the autograd engine's device thread while a backward runs
        # (the main thread then sits in run_backward), else the main thread
        f = frames.get(main_id)
        if f is not None and f.f_code.co_name in ('_engine_run_backward', 'backward'):
            for tid, g in frames.items():
                if tid not in (me, main_id):
                    f = g
                    break
        if f is not None:
            inner['%s:%d %s' % (os.path.relpath(f.f_code.co_filename, ROOT)[-50:], f.f_lineno, f.f_code.co_name)] += 1
            g, chain = f, []
            while g is not None:
                fn = g.f_code.co_filename
                if fn.startswith(ROOT) and 'tools/host_sampler' not in fn:
                    chain.append('%s:%d %s' % (os.path.relpath(fn, ROOT), g.f_lineno, g.f_code.co_name))
                g = g.f_back
            if chain:
                ours[chain[0]] += 1
                phases[chain[-1] if len(chain) < 2 else chain[-2]] += 1
        time.sleep(0.0001)


sys.setswitchinterval(0.0002)
th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.perf_counter()
for _ in range(STEPS):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
stop[0] = True
th.join()
n = sum(inner.values())
print('per step: host %.2f ms, wall %.2f ms; %d samples (%.0f per step), median sampling gap %.3f ms' %
      (host / STEPS * 1e3, wall / STEPS * 1e3, n, n / STEPS, sorted(gaps)[len(gaps) // 2] * 1e3))
print('\ninnermost frame (any file), share of samples:')
for k, v in inner.most_common(25):
    print('  %5.1f %%  %s' % (100.0 * v / n, k))
print('\ninnermost frame inside this repository:')
for k, v in ours.most_common(40):
    print('  %5.1f %%  %s' % (100.0 * v / n, k))
print('\nsecond-outermost repository frame (phase):')
for k, v in phases.most_common(15):
    print('  %5.1f %%  %s' % (100.0 * v / n, k))
