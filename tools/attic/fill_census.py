"""Which call sites issue the ~380 fill kernels of a KD step (torch.zeros / new_zeros / zeros_like / zero_ / fill_ / full /
ones): count and bytes per step by the innermost frame inside this repository."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from u2mkd_amd import lidar, train as T, kd as KD
from u2mkd_amd.synth import synth_kd_batch

sites = collections.defaultdict(lambda: [0, 0])
active = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and 'fill_census' not in fr.filename:
            return '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
    return 'torch-internal'


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if active[0] and torch.is_tensor(out) and out.is_cuda:
            s = sites[name + ' @ ' + site()]
            s[0] += 1; s[1] += out.numel() * out.element_size()
        return out
    setattr(owner, name, f)


for owner, names in ((torch, ('zeros', 'zeros_like', 'full', 'ones', 'ones_like', 'full_like')),
                     (torch.Tensor, ('new_zeros', 'zero_', 'fill_', 'new_full', 'new_ones'))):
    for n in names:
        wrap(owner, n)

torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
run = T.KDStep(model, num_epochs=50, batch_size=1)
run.train_mode()
res = [T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234 + i, image_hw=(360, 640))) for i in range(2)]
for i in range(3):
    run(T.fresh_batch(res[i % 2]))
torch.cuda.synchronize()
active[0] = True
STEPS = 2
for i in range(STEPS):
    run(T.fresh_batch(res[i % 2]))
torch.cuda.synchronize()
active[0] = False
tot = sum(v[0] for v in sites.values())
print('python-level fills per step: %.0f' % (tot / STEPS))
for k, v in sorted(sites.items(), key=lambda kv: -kv[1][0])[:45]:
    print('%6.1f /step %9.2f MB/step  %s' % (v[0] / STEPS, v[1] / STEPS / 1e6, k))
