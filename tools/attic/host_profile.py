"""cProfile of the host side of the SPVCNN training step (is the step launch-bound?)."""
import sys, cProfile, pstats, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import lidar, train as T
from u2mkd_amd.synth import synth_batch
b = synth_batch(80000, 1)
feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
run = T.LidarStep(model)
for _ in range(3): run(feats, coords, labels)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): run(feats, coords, labels)
t_launch = time.perf_counter() - t
torch.cuda.synchronize()
print('host launch time per step %.1f ms, wall per step %.1f ms' % (t_launch / 5 * 1e3, (time.perf_counter() - t) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): run(feats, coords, labels)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
