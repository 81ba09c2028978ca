import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, numpy as np
import torch.nn.functional as F
from oracle.spvcnn_ref import fill_state_by_name
from u2mkd_amd import kd, lidar, pixel_head, torchsparse as ts
from u2mkd_amd.synth import synth_kd_batch
from u2mkd_amd.fusion import feature_fetch
from test_gpu_pixel_head import _dense_fp64
b = synth_kd_batch(1200, 2, seed=78, image_hw=(64, 112)); s = b['student']
sp = {k: v for k, v in lidar.spformer_kwargs(drop_path_rate=0.0).items() if k not in ('cr', 'in_channel', 'num_classes')}
model = fill_state_by_name(kd.TSDFull(cr=1.0, cr_t=1.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp)).cuda().train()
ms_ = model.model_s; ms_.dropout.p = 0.0
pc = [torch.from_numpy(c).cuda() for c in s['pixel_coordinates']]; ms = [torch.from_numpy(m).cuda() for m in s['masks']]
cap = {}
orig = pixel_head.sampled_pixel_logits
def grab(x, head, *a):
    cap['x'] = x.detach().clone(); return orig(x, head, *a)
kd.sampled_pixel_logits = grab
stu = {'lidar': ts.SparseTensor(torch.from_numpy(s['feats']).cuda(), torch.from_numpy(s['coords']).cuda()),
       'images': torch.from_numpy(s['images']).permute(0, 1, 4, 2, 3).contiguous().cuda(), 'pixel_coordinates': pc, 'masks': ms, 'fov_mask': torch.from_numpy(s['fov_mask']).cuda()}
with torch.no_grad(): out = ms_(stu)
x = cap['x']; head = ms_.classifier_pix
print('x', tuple(x.shape), 'abs max %.3g' % float(x.abs().max()))
mean = x.mean((0,2,3)); std = x.std((0,2,3)); k0 = x[0,:,0,0]
print('per-channel mean/std ratio: max %.3g median %.3g; |mean-K|/std max %.3g' % (float((mean.abs()/std).max()), float((mean.abs()/std).median()), float(((mean-k0).abs()/std).max())))
with torch.no_grad():
    got = orig(x, head, pc, ms, (64,112), 2, 6)
    want, u = _dense_fp64(x.double(), head, pc, ms, (64,112), 2, 6)
    fm = head(F.interpolate(x, (64,112), mode='bilinear', align_corners=True))
    dense32 = feature_fetch(ms, pc, fm.view(2,6,17,64,112))
sc = float(want.abs().max())
print('sampled vs fp64 %.3g ; dense-fp32 vs fp64 %.3g ; scale %.3g' % (float((got.double()-want).abs().max())/sc, float((dense32.double()-want).abs().max())/sc, sc))
# whole student, sampled head on / off, same model, same process
kd.sampled_pixel_logits = orig
res = {}
for rep in range(2):
    for en in (True, False):
        pixel_head._ENABLED = en
        with torch.no_grad():
            o = ms_(dict(stu))
        torch.cuda.synchronize()
        res[(rep, en)] = o['x_pix'].clone()
sc = float(res[(0, False)].abs().max())
print('dense run0 vs run1 %.3g | sampled run0 vs run1 %.3g | sampled vs dense %.3g (scale %.3g)' % (
    float((res[(0, False)] - res[(1, False)]).abs().max()) / sc, float((res[(0, True)] - res[(1, True)]).abs().max()) / sc,
    float((res[(0, True)] - res[(0, False)]).abs().max()) / sc, sc))
