"""Where the host waits during a KD step: time inside every blocking Tensor.item() / .tolist() / .cpu() / nonzero
call by call site (ms per step), plus the host-side marks of the step (teacher issued, student forward issued,
backward + optimizer returned, GPU drained).  Env knobs of DESIGN.md section 7a apply."""
import collections, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import kd as KD, lidar, train as T
from u2mkd_amd.synth import synth_kd_batch

torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
run = T.KDStep(model, num_epochs=50, batch_size=1)
run.train_mode()
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=(360, 640)))
marks = collections.defaultdict(list)
t0 = [0.0]


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        marks[key + ' begin'].append((time.perf_counter() - t0[0]) * 1e3)
        r = f(*a, **k)
        marks[key + ' end'].append((time.perf_counter() - t0[0]) * 1e3)
        return r
    setattr(obj, name, g)


wrap(model.model_t, 'forward', 'teacher forward')
wrap(model.model_s, 'forward', 'student forward')
wrap(run.amp, 'backward_and_step', 'backward+optimizer')
waits, counts = collections.Counter(), collections.Counter()
for meth in ('item', 'tolist', 'cpu'):
    orig = getattr(torch.Tensor, meth)

    def timed(self, *a, _orig=orig, _m=meth, **k):
        if not self.is_cuda:
            return _orig(self, *a, **k)
        t = time.perf_counter()
        r = _orig(self, *a, **k)
        dt = time.perf_counter() - t
        st = [f for f in traceback.extract_stack() if 'u2mkd_amd' in f.filename]
        key = _m + ' @ ' + ' <- '.join('%s:%d' % (f.filename.split('/')[-1], f.lineno) for f in st[-2:][::-1])
        waits[key] += dt
        counts[key] += 1
        return r
    setattr(torch.Tensor, meth, timed)
_orig_unique = torch.unique


def _timed_unique(*a, **k):
    t = time.perf_counter()
    r = _orig_unique(*a, **k)
    n = r[0].shape[0] if isinstance(r, tuple) else r.shape[0]       # the size is known: the stream has been synchronised
    dt = time.perf_counter() - t
    st = [f for f in traceback.extract_stack() if 'u2mkd_amd' in f.filename]
    key = 'unique @ ' + ' <- '.join('%s:%d' % (f.filename.split('/')[-1], f.lineno) for f in st[-2:][::-1]) + \
        ' [stream %s]' % ('main' if torch.cuda.current_stream() == torch.cuda.default_stream() else 'side')
    waits[key] += dt
    counts[key] += 1
    return r


torch.unique = _timed_unique
for _ in range(6):
    run(d)
torch.cuda.synchronize()
marks.clear(); waits.clear(); counts.clear()
K = 12
t_all = time.perf_counter()
for _ in range(K):
    t0[0] = time.perf_counter()
    run(d)
    marks['step returned'].append((time.perf_counter() - t0[0]) * 1e3)
torch.cuda.synchronize()
print('wall %.1f ms/step (no synchronisation between steps)' % ((time.perf_counter() - t_all) / K * 1e3))
for k, v in marks.items():
    print('  %-28s %7.2f ms' % (k, sum(v) / len(v)))
print('host waits by call site (ms per step):')
for k, v in waits.most_common(12):
    print('  %7.2f ms %5.1f calls  %s' % (v / K * 1e3, counts[k] / K, k))

if os.environ.get('U2MKD_PROBE_TAIL') == '1':
    # while the tail of a step's backward is still running on the main stream: how long does a trivial kernel + sync
    # take on (a) the teacher's stream, (b) a stream nothing has used in this step, (c) after a 2 ms pause
    fresh = torch.cuda.Stream()

    class bench_pin:
        pass
    for name in ('teacher', 'fresh', 'teacher'):
        torch.cuda.synchronize()
        run(d)                                      # returns with ~30 ms of GPU work still queued
        s = KD._SIDE[(0, 'teacher')] if name == 'teacher' else fresh
        with torch.cuda.stream(s):
            t = time.perf_counter()
            x = torch.zeros(8, device='cuda'); x.add_(1); ev = torch.cuda.Event(); ev.record(); ev.synchronize()
            dt0 = (time.perf_counter() - t) * 1e3
            t = time.perf_counter()
            pin = torch.empty(1, dtype=torch.float32).pin_memory() if not hasattr(bench_pin, 'p') else bench_pin.p
            bench_pin.p = pin
            pin.copy_(x.sum().view(1), non_blocking=True); ev2 = torch.cuda.Event(); ev2.record(); ev2.synchronize()
            dtp = (time.perf_counter() - t) * 1e3
            t = time.perf_counter()
            x = torch.zeros(8, device='cuda'); x.add_(1); v = x.sum().item()
            dt1 = (time.perf_counter() - t) * 1e3
            t = time.perf_counter()
            y = torch.arange(80000, device='cuda') % 977; u = _orig_unique(y).shape[0]
            dt2 = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); torch.cuda.synchronize(); rest = (time.perf_counter() - t) * 1e3
        print('%-8s stream during the tail: tiny kernel + event sync %.2f ms, + pinned copy + event %.2f ms, + .item() %.2f ms, unique(80k) %.2f ms; the tail then took %.1f ms more' % (name, dt0, dtp, dt1, dt2, rest))
