"""cProfile of the host side of the SPVCNN step on a 2000-voxel scene (GPU work ~0)."""
import sys, cProfile, pstats, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import lidar, train as T
from u2mkd_amd.synth import synth_batch
b = synth_batch(2000, 1)
feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
run = T.LidarStep(model)
for _ in range(5): run(feats, coords, labels)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run(feats, coords, labels)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
