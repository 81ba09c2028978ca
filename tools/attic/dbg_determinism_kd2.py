"""Per-module output checksums of the student over repeated identical KD steps: which module first differs by more than
rounding noise?  python tools/dbg_determinism_kd2.py [n_pts] [runs]"""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nb = synth_kd_batch(n, 1, seed=1234, image_hw=(64, 112), sweeps=1)
d = T.kd_batch_to_device(nb)
run = _runner(2.0, 2.0, amp=False)
logs = []


def hook(name):
    def f(mod, inp, out):
        t = out.F if hasattr(out, 'F') else out
        if isinstance(t, (tuple, list)):
            t = t[0]
        if torch.is_tensor(t) and t.is_floating_point():
            x = t.detach().double()
            logs[-1].append((name, torch.stack([x.sum(), x.abs().sum()])))
    return f


from u2mkd_amd.torchsparse.nn import functional as F
import u2mkd_amd.lidar.point_voxel as PV


def cs(t):
    x = t.detach().double()
    return torch.stack([x.sum(), x.abs().sum()])


def wrap(mod, fname, tag):
    real = getattr(mod, fname)

    def f(*a, **k):
        out = real(*a, **k)
        if torch.cuda.current_stream() == torch.cuda.default_stream() and torch.is_grad_enabled():
            ins = [cs(x) for x in a if torch.is_tensor(x) and x.is_floating_point()]
            logs[-1].append((tag + ':in', torch.cat(ins) if ins else torch.zeros(2, dtype=torch.double, device='cuda')))
            logs[-1].append((tag + ':out', cs(out)))
        return out
    setattr(mod, fname, f)


def wrap_int(mod, fname, tag):
    real = getattr(mod, fname)

    def f(*a, **k):
        out = real(*a, **k)
        if torch.cuda.current_stream() == torch.cuda.default_stream() and torch.is_grad_enabled():
            outs_ = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs_):
                if torch.is_tensor(o):
                    x = o.detach().double()
                    w = torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.double).view(x.shape) % 1009
                    logs[-1].append(('%s:out%d' % (tag, i), torch.stack([x.sum(), (x * w).sum()])))
        return out
    setattr(mod, fname, f)


wrap_int(F, 'ti_weights_n8', 'ti_weights')
wrap_int(F.HashTable, 'query', 'hash_query')
wrap_int(F, 'sphash', 'sphash')
wrap_int(F, 'spcount', 'spcount')
wrap(F, '_conv_os', 'conv_os')
wrap(F, 'spvoxelize', 'spvoxelize')
wrap(F, 'spdevoxelize', 'spdevoxelize')
wrap(F, 'batch_norm', 'batch_norm')
for m_ in (PV,):
    for fn in ('spvoxelize', 'spdevoxelize'):
        if hasattr(m_, fn):
            setattr(m_, fn, getattr(F, fn))
for name, mod in run.model.model_s.named_modules():
    if name:
        mod.register_forward_hook(hook(name))
outs = []
for r in range(runs):
    logs.append([])
    out, ld = _step(run, d, False)
    outs.append(out['stu']['x_vox'].detach().clone())
torch.cuda.synchronize()
for r in range(1, runs):
    dx = float((outs[r] - outs[r - 1]).abs().max())
    diffs = []
    for (na, a), (nb_, b) in zip(logs[r], logs[r - 1]):
        if a.shape == b.shape and not torch.equal(a, b):
            diffs.append((na, float(((a - b).abs() / (b.abs() + 1e-30)).max())))
    print('run %d vs %d: logits max diff %.3g | %d of %d modules differ; first five: %s' % (r, r - 1, dx, len(diffs), len(logs[r]), diffs[:5]))
