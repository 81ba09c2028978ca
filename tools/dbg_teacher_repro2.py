"""Which teacher module is the first whose output differs when a KD step's teacher logits deviate (tools/dbg_teacher_repro.py):
forward hooks keep a float64 checksum of every leaf module's output; a deviating step lists the modules whose checksum differs
from the first step's, in execution order.   python tools/dbg_teacher_repro2.py [steps=40] [H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (900, 1600)
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=hw))
run = _runner(1.0, 2.0)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
log = []


def feats_of(o):
    if torch.is_tensor(o):
        return o
    if hasattr(o, 'F') and torch.is_tensor(o.F):
        return o.F
    if isinstance(o, dict):
        for v in o.values():
            f = feats_of(v)
            if f is not None:
                return f
    if isinstance(o, (list, tuple)):
        for v in o:
            f = feats_of(v)
            if f is not None:
                return f
    return None


def hook(name):
    def f(mod, inp, out):
        t = feats_of(out)
        if t is not None and t.is_floating_point():
            log.append((name, t.detach().double().sum().reshape(1), t.detach().double().abs().sum().reshape(1)))
    return f


for name, mod in run.model.model_t.named_modules():
    if name and not list(mod.children()):
        mod.register_forward_hook(hook(name))
ref = None
for i in range(steps):
    run.model.load_state_dict(state)
    log.clear()
    out, ld = _step(run, d, False)
    names = [n for n, _, _ in log]
    vals = torch.cat([torch.cat([a, b]) for _, a, b in log]).cpu()
    if ref is None:
        ref = (names, vals)
        print('%d leaf-module outputs per teacher forward' % len(names), flush=True)
        continue
    assert names == ref[0]
    bad = ((vals != ref[1]).view(-1, 2).any(1)).nonzero().view(-1).tolist()
    if bad:
        print('step %d: %d modules differ; first: %s' % (i, len(bad), [names[j] for j in bad[:6]]), flush=True)
print('done', flush=True)
