"""Which side is noisy?  GPU fp32 and CPU-oracle fp32 gradients vs a CPU fp64 run."""
import sys; sys.path.insert(0, '.')
import torch
from oracle import spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd.synth import synth_batch
from u2mkd_amd.losses import MixLovaszCrossEntropy
n, bsz, cr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
b = synth_batch(n, bsz, 11)
feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))
kw = dict(cr=cr, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
crit = MixLovaszCrossEntropy(ignore_index=0)
def run_cpu(dtype):
    m = O.fill_state_by_name(O.SPVCNN(**kw)).train().to(dtype); m.dropout.p = 0.0
    out = m({'lidar': ots.SparseTensor(feats.to(dtype), coords)})['x_vox']
    out.retain_grad()
    crit(out, labels).backward()
    return m, out
m64, o64 = run_cpu(torch.float64)
m32, o32 = run_cpu(torch.float32)
from u2mkd_amd import lidar, torchsparse as ts
mg = lidar.SPVCNN(**kw); mg.load_state_dict(m32.state_dict()); mg.cuda().train(); mg.dropout.p = 0.0
og = mg({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox']
og.retain_grad()
crit(og, labels.cuda()).backward()
print('logits: cpu32 vs 64 %.2e   gpu32 vs 64 %.2e' % (float((o32.double()-o64).abs().max()), float((og.detach().cpu().double()-o64).abs().max())))
s = float(o64.grad.abs().max())
print('dlogits: cpu32 %.2e gpu32 %.2e' % (float((o32.grad.double()-o64.grad).abs().max())/s, float((og.grad.cpu().double()-o64.grad).abs().max())/s))
# loss grad with IDENTICAL logits on both devices
x = o32.detach().clone().requires_grad_(True); crit(x, labels).backward()
xg = o32.detach().clone().cuda().requires_grad_(True); crit(xg, labels.cuda()).backward()
print('loss-only grad gpu vs cpu (same logits): %.2e' % (float((xg.grad.cpu()-x.grad).abs().max())/float(x.grad.abs().max())))
g64 = dict(m64.named_parameters()); g32 = dict(m32.named_parameters())
names = [n_ for n_, _ in mg.named_parameters()]
for name, p in mg.named_parameters():
    if not (name.startswith('vox_ups.3') or name.startswith('classifier') or name.startswith('point_transforms.2')): continue
    r = g64[name].grad; s = float(r.abs().max())
    if s < 1e-9: continue
    print(f'{name:42s} cpu32 {float((g32[name].grad.double()-r).abs().max())/s:.2e}  gpu32 {float((p.grad.cpu().double()-r).abs().max())/s:.2e}')
