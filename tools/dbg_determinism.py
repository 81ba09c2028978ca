"""Which module of the frozen teacher first differs between two identical forwards?  (python tools/dbg_determinism.py [f32|bf16] [n_pts])"""
import sys
import torch
sys.path.insert(0, '.')
from u2mkd_amd import lidar, torchsparse as ts
from u2mkd_amd.synth import synth_batch

amp = (sys.argv[1] if len(sys.argv) > 1 else 'bf16') == 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
b = synth_batch(n, 1, seed=1234, sweeps=9)
feats, coords = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords'))
torch.manual_seed(0)
m = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=2.0)).cuda().eval()
logs = []


def hook(name):
    def f(mod, inp, out):
        t = out.F if hasattr(out, 'F') else out
        if torch.is_tensor(t):
            logs[-1].append((name, t.detach().float().double().sum().item(), t.detach().float().abs().double().sum().item()))
    return f


for name, mod in m.named_modules():
    if name:
        mod.register_forward_hook(hook(name))
outs = []
for r in range(2):
    logs.append([])
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        outs.append(m({'lidar': ts.SparseTensor(feats, coords)})['x_vox'].clone())
torch.cuda.synchronize()
print('equal outputs:', torch.equal(outs[0], outs[1]), 'amp', amp, 'n', n)
bad = 0
for a, c in zip(logs[0], logs[1]):
    if a != c:
        print('DIFF', a[0], a[1:], c[1:])
        bad += 1
        if bad > 12:
            break
print('modules compared', len(logs[0]), 'first diffs shown', bad)
