"""HIP vs CPU-oracle logits of the multi-sweep teacher fixture recipe over a few seeds (fixture seed selection)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import spformer_ref as R, spvcnn_ref as O, torchsparse_cpu as ots
from u2mkd_amd import lidar, torchsparse as ts
from u2mkd_amd.synth import synth_batch
torch.set_num_threads(16)
for seed in range(55, 63):
    b = synth_batch(3000, 2, seed=seed, sweeps=3)
    feats, coords = torch.from_numpy(b['feats']), torch.from_numpy(b['coords'])
    ref = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))).train()
    ref.dropout.p = 0.0
    with torch.no_grad():
        want = ref({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
    m = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0, drop_path_rate=0.0))
    m.load_state_dict(ref.state_dict()); m.cuda().train(); m.dropout.p = 0.0
    with torch.no_grad():
        out = m({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox'].cpu()
    d = (out - want).abs()
    r = int(d.max(1).values.argmax())
    print('seed', seed, 'max %.3g (row max |logit| %.3g) rows>1e-3: %d rows>5e-4: %d median %.3g logit range %.3g'
          % (float(d.max()), float(want[r].abs().max()), int((d.max(1).values > 1e-3).sum()), int((d.max(1).values > 5e-4).sum()),
             float(d.median()), float(want.abs().max())), flush=True)
