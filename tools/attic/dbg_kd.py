import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from tests.test_kd_path import _build, _kd_tensors, G
from oracle.spvcnn_ref import fill_state_by_name
from u2mkd_amd import kd, torchsparse as ts
from u2mkd_amd.synth import synth_kd_batch
gold = np.load(os.path.join(G, 'kd_cr10_3000_edge_seed77.npz'))
model = fill_state_by_name(_build('cuda')).cuda().train(); model.model_t.eval(); model.model_s.dropout.p = 0.0
b = synth_kd_batch(1500, 2, seed=77, image_hw=(64, 112)); s, t = b['student'], b['teacher']
pc, ms = _kd_tensors(b, 'cuda')
stu = {'lidar': ts.SparseTensor(torch.from_numpy(s['feats']).cuda(), torch.from_numpy(s['coords']).cuda()),
       'images': torch.from_numpy(s['images']).permute(0, 1, 4, 2, 3).contiguous().cuda(),
       'pixel_coordinates': pc, 'masks': ms, 'fov_mask': torch.from_numpy(s['fov_mask']).cuda()}
tea = {'lidar': ts.SparseTensor(torch.from_numpy(t['feats']).cuda(), torch.from_numpy(t['coords']).cuda())}
with torch.no_grad():
    out = model({'student': stu, 'teacher': tea})
for name, a, key in (('x_vox_t', out['t']['x_vox'], 'x_vox_t'), ('x_vox', out['stu']['x_vox'], 'x_vox'), ('x_pix', out['stu']['x_pix'], 'x_pix'), ('pts_feats', out['stu']['pts_feats'][0][::16], 'pts_feats_s')):
    e = (a.cpu() - torch.from_numpy(gold[key])).abs()
    rows = e.max(1)[0]
    print(f'{name:10s} max {float(e.max()):.2e} median {float(e.median()):.2e} rows>1e-3: {int((rows > 1e-3).sum())}/{len(rows)} p99.9 {float(torch.quantile(e.flatten()[:1000000], 0.999)):.2e}')
print('mse', [float(m) for m in out['stu']['mse_loss']], gold['mse'].tolist())
