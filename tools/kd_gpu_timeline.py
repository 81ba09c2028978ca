"""Un-profiled GPU timeline of one KD step: HIP events recorded (on whatever stream is current) when selected modules
finish their forward, and around the step's phases; printed relative to the step's first event.
python tools/kd_gpu_timeline.py"""
import sys, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import kd as KD, train as T
from tools.kd_host import build

run, d0 = build(80000)
events = []          # (label, event, host time)
on = [False]


def mark(label):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        events.append((label, e, time.perf_counter()))


def hook(label):
    def f(mod, inp, out):
        mark(label)
    return f


ms, mt = run.model.model_s, run.model.model_t
for name, mod in [('S.stem', ms.stem)] + [('S.down%d' % i, m) for i, m in enumerate(ms.vox_downs)] + \
        [('S.attn%d' % i, m) for i, m in enumerate(ms.transformer_blocks)] + [('S.c2l%d' % i, m) for i, m in enumerate(ms.c2l_fusion_blocks)] + \
        [('S.up%d' % i, m) for i, m in enumerate(ms.vox_ups)] + [('S.cls', ms.classifier_vox)] + \
        [('T.stem', mt.stem)] + [('T.down%d' % i, m) for i, m in enumerate(mt.vox_downs)] + [('T.attn%d' % i, m) for i, m in enumerate(mt.transformer_blocks)] + \
        [('T.up%d' % i, m) for i, m in enumerate(mt.vox_ups)] + [('T.cls', mt.classifier_vox)] + \
        [('C.layer%d' % i, getattr(ms.pix_branch, 'layer%d' % i)) for i in (1, 2, 3, 4)]:
    mod.register_forward_hook(hook(name))
orig_prepare, orig_bwd, orig_loss = run.model.prepare, run.amp.backward_and_step, KD.kd_losses


def prep(*a, **k):
    mark('prepare.begin'); o = orig_prepare(*a, **k); mark('prepare.end'); return o


def bwd(*a, **k):
    mark('backward.begin'); o = orig_bwd(*a, **k); mark('step.end'); return o


def loss(*a, **k):
    mark('losses.begin'); o = orig_loss(*a, **k); mark('losses.end'); return o


run.model.prepare, run.amp.backward_and_step, KD.kd_losses = prep, bwd, loss
cur = T.fresh_batch(d0)
for i in range(9):
    if i == 7:
        torch.cuda.synchronize()
        on[0] = True
        mark('step.begin')
    if i == 8:
        on[0] = False
    nxt = T.fresh_batch(d0)
    run(cur, prefetch=nxt)
    cur = nxt
torch.cuda.synchronize()
e0, h0 = events[0][1], events[0][2]
print('%-16s %9s %9s' % ('event', 'GPU ms', 'host ms'))
for label, e, h in events:
    print('%-16s %9.2f %9.2f' % (label, e0.elapsed_time(e), (h - h0) * 1e3))
