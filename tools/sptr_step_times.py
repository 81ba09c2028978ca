"""Duration of every sptr attention launch of one KD step (student + teacher, forward and backward), by HIP events
around the C-ABI calls; also the window statistics of every call (tokens, mean / max window length).
Run once per U2MKD_SPTR_SPLIT value for an A/B of the key-split kernels.   python tools/sptr_step_times.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import _lib as L, train as T
from tools.kd_host import build

run, d0 = build(80000)
rec, on = [], [False]
orig = L.call


def call(name, *a):
    if on[0] and name.startswith('u2mkd_sptr_attention'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream()); orig(name, *a); e1.record(torch.cuda.current_stream())
        rec.append((name.replace('u2mkd_sptr_attention_', ''), e0, e1, torch.cuda.current_stream().cuda_stream))
    else:
        orig(name, *a)


L.call = call
import u2mkd_amd.sptr.functional as SF
SF.L.call = call
stats = []
oi = SF.WindowPlan.__init__


def init(self, *a, **k):
    oi(self, *a, **k)
    if on[0]:
        w = self.wlen.float()
        stats.append((int(self.n), float(w.mean()), int(w.max()), float((w * w).sum())))


SF.WindowPlan.__init__ = init
cur = T.fresh_batch(d0)
for i in range(5):
    on[0] = i == 4
    nxt = T.fresh_batch(d0)
    run(cur)
    cur = nxt
torch.cuda.synchronize()
print('split', os.environ.get('U2MKD_SPTR_SPLIT', 'default'))
tot = {}
for name, e0, e1, st in rec:
    ms = e0.elapsed_time(e1)
    tot[name.split('_')[0]] = tot.get(name.split('_')[0], 0) + ms
    print('%-18s stream %x  %8.3f ms' % (name, st & 0xffff, ms))
print('totals', {k: round(v, 3) for k, v in tot.items()})
for s in stats:
    print('window plan: tokens %6d  mean window %7.1f  max %5d  pairs %.3g' % s)
