"""Fixture seed selection for the first KD golden (cr 1.0 / cr_t 1.0, 2 x 1500 voxels): for every candidate seed the
HIP model is evaluated twice on the GPU box -- spherical coordinates from the GPU's libm and from the CPU's (same fp32
formula) -- and the largest change of the student's / teacher's outputs is printed.  A seed whose outputs do not move
holds no token within the two libms' last-place difference of an edge of SphereFormer's hard quantisers; make_golden.py
then asserts the wider +-4 ulp margin on the CPU before it stores the fixture."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle.spvcnn_ref import fill_state_by_name
from u2mkd_amd import kd, lidar, torchsparse as ts
from u2mkd_amd.lidar import sphereformer as SFM
from u2mkd_amd.synth import synth_kd_batch

seeds = [int(a) for a in sys.argv[1:]] or list(range(77, 93))
sp = {k: v for k, v in lidar.spformer_kwargs(drop_path_rate=0.0).items() if k not in ('cr', 'in_channel', 'num_classes')}
model = fill_state_by_name(kd.TSDFull(cr=1.0, cr_t=1.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp)).cuda().train()
model.model_t.eval()
model.model_s.dropout.p = 0.0
gpu_fn = SFM.cart2sphere


def forward(b):
    s, t = b['student'], b['teacher']
    pc = [torch.from_numpy(c).cuda() for c in s['pixel_coordinates']]
    ms = [torch.from_numpy(m).cuda() for m in s['masks']]
    stu = {'lidar': ts.SparseTensor(torch.from_numpy(s['feats']).cuda(), torch.from_numpy(s['coords']).cuda()),
           'images': torch.from_numpy(s['images']).permute(0, 1, 4, 2, 3).contiguous().cuda(),
           'pixel_coordinates': pc, 'masks': ms, 'fov_mask': torch.from_numpy(s['fov_mask']).cuda()}
    tea = {'lidar': ts.SparseTensor(torch.from_numpy(t['feats']).cuda(), torch.from_numpy(t['coords']).cuda())}
    with torch.no_grad():
        out = model({'student': stu, 'teacher': tea})
    return out['stu']['x_vox'].clone(), out['stu']['pts_feats'][0].clone(), out['t']['x_vox'].clone(), out['stu']['x_pix'].clone()


for seed in seeds:
    b = synth_kd_batch(1500, 2, seed=seed, image_hw=(64, 112))
    SFM.cart2sphere = gpu_fn
    a = forward(b)
    a2 = forward(b)
    SFM.cart2sphere = lambda xyz: gpu_fn(xyz.detach().cpu()).to(xyz.device)
    c = forward(b)
    SFM.cart2sphere = gpu_fn
    d = [float((x - y).abs().max()) for x, y in zip(a, c)]
    r = [float((x - y).abs().max()) for x, y in zip(a, a2)]
    print('seed %d  gpu-libm vs cpu-libm angles: x_vox %.3g pts_feats %.3g x_vox_t %.3g x_pix %.3g | run to run %.3g %.3g %.3g %.3g'
          % (seed, *d, *r), flush=True)
