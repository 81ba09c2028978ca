"""Host-side timeline of the KD step with the geometry prefetch: how long the Python thread spends in each phase (launch
issue incl. any wait inside), and how far behind the GPU is when the host leaves the step.
python tools/kd_phases.py [steps] [noprefetch]"""
import sys, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import kd as KD, train as T
from tools.kd_host import build

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prefetch = not (len(sys.argv) > 2 and sys.argv[2] == 'noprefetch')
run, d0 = build(80000)
acc = {}


def timed(name, fn):
    def f(*a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    return f


run.model.model_t.forward = timed('teacher_fwd', run.model.model_t.forward)
run.model.model_s.forward = timed('student_fwd', run.model.model_s.forward)
run.model.prepare = timed('prepare_next', run.model.prepare)
run.amp.backward_and_step = timed('backward+opt', run.amp.backward_and_step)
KD.kd_losses = timed('losses', KD.kd_losses)


def loop(n):
    cur = T.fresh_batch(d0)
    for _ in range(n):
        nxt = T.fresh_batch(d0) if prefetch else None
        run(cur, prefetch=nxt)
        cur = nxt if prefetch else T.fresh_batch(d0)


loop(5)
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
loop(steps)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print('prefetch', prefetch, '| per step (ms): ' + '  '.join('%s %.1f' % (k, 1e3 * v / steps) for k, v in acc.items()))
print('wall %.1f ms/step; host returned after %.1f ms/step; final drain %.1f ms' % (1e3 * wall / steps, 1e3 * t_host / steps, 1e3 * (wall - t_host)))
