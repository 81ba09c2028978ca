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
        [('S.up%d.%d' % (i, j), mm) for i, m in enumerate(ms.vox_ups) for j, mm in enumerate(m)] + [('S.cls', ms.classifier_vox)] + \
        [('S.pt%d' % i, m) for i, m in enumerate(ms.point_transforms)] + \
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
# backward: events when the gradient with respect to a LiDAR module's input is ready (camera pieces replay from graphs)
if '--bwd' in sys.argv:
    def bhook(label):
        def f(mod, gin, gout):
            mark(label)
        return f
    for name, mod in [('bS.stem', ms.stem)] + [('bS.down%d' % i, m) for i, m in enumerate(ms.vox_downs)] + \
            [('bS.attn%d' % i, m) for i, m in enumerate(ms.transformer_blocks)] + \
            [('bS.up%d.%d' % (i, j), mm) for i, m in enumerate(ms.vox_ups) for j, mm in enumerate(m)] + \
            [('bS.pt%d' % i, m) for i, m in enumerate(ms.point_transforms)] + [('bS.cls', ms.classifier_vox)] + \
            [('bS.c2l%d' % i, m) for i, m in enumerate(ms.c2l_fusion_blocks)]:
        mod.register_full_backward_hook(bhook(name))
# camera pieces (hipGraph replays): begin / end of every forward and backward replay on the stream that runs it
from u2mkd_amd import graphs as GR
_call, _rb = GR.StaticPiece.__call__, GR._Replay.backward


def piece_call(self, *a):
    mark('G.%s.begin' % self.name); o = _call(self, *a); mark('G.%s.end' % self.name); return o


def replay_bwd(ctx, *g):
    mark('bG.%d.begin' % id(ctx.rec)); o = _rb(ctx, *g); mark('bG.%d.end' % id(ctx.rec)); return o


GR.StaticPiece.__call__ = piece_call
GR._Replay.backward = staticmethod(replay_bwd)
STEADY = '--steady' in sys.argv       # no synchronisation before the observed steps: the pipelined steady state
cur = T.fresh_batch(d0)
for i in range(10):
    if i == 7:
        if not STEADY:
            torch.cuda.synchronize()
        on[0] = True
        mark('step.begin')
    if i == (9 if STEADY else 8):
        on[0] = False
    elif i == 8:
        mark('step.begin')
    nxt = T.fresh_batch(d0)
    run(cur, prefetch=nxt)
    cur = nxt
torch.cuda.synchronize()
names = {id(r): pc.name for pc in GR._LIVE for r in pc._records.values() if r is not None}
events = [(('bG.%s.%s' % (names.get(int(l.split('.')[1]), '?'), l.split('.')[2])) if l.startswith('bG.') else l, e, h) for l, e, h in events]
e0, h0 = events[0][1], events[0][2]
print('%-16s %9s %9s' % ('event', 'GPU ms', 'host ms'))
for label, e, h in events:
    print('%-16s %9.2f %9.2f' % (label, e0.elapsed_time(e), (h - h0) * 1e3))
