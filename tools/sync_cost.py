"""Where the host waits during a KD step: time inside every blocking Tensor.item() / .tolist() / .cpu() / nonzero
call by call site (ms per step), plus the host-side marks of the step (teacher issued, student forward issued,
backward + optimizer returned, GPU drained).  Env knobs apply (e.g. U2MKD_TEACHER_AHEAD=1)."""
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
