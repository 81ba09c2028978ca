import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_golden_teacher_multisweep import _hip_model, _inputs
from u2mkd_amd import torchsparse as ts
feats, coords, labels, kf = (t.cuda() for t in _inputs())
m = _hip_model()
outs = {}
logs = {}
def hook(tag):
    def f(name):
        def g(mod, inp, out):
            t = out.F if hasattr(out, 'F') else out
            if isinstance(t, (tuple, list)): t = t[0]
            if torch.is_tensor(t) and t.is_floating_point():
                logs.setdefault(tag, []).append((name, t.detach().clone()))
        return g
    return f
for mode in ('grad', 'nograd', 'grad2'):
    hs = [mod.register_forward_hook(hook(mode)(n)) for n, mod in m.named_modules() if n]
    with torch.set_grad_enabled(mode != 'nograd'):
        outs[mode] = m({'lidar': ts.SparseTensor(feats, coords)})['x_vox'].detach().clone()
    for h in hs: h.remove()
print('grad vs nograd max', float((outs['grad'] - outs['nograd']).abs().max()), 'grad vs grad2', float((outs['grad'] - outs['grad2']).abs().max()))
for (n, a), (_, b) in zip(logs['grad'], logs['nograd']):
    if a.shape == b.shape:
        d = float((a - b).abs().max())
        if d > 0:
            print('first differing module:', n, d, 'scale', float(b.abs().max()))
            break
