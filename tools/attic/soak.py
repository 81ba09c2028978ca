"""Soak run: full SPVCNN training steps over random scene sizes / batch sizes (dynamic shapes: schedule
caches, grow-only scratch, plan caches) -- every loss finite, memory bounded."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import lidar, train as T
from u2mkd_amd.synth import synth_batch
rng = np.random.default_rng(0)
model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
run = T.LidarStep(model)
peak0 = None
t0 = time.time()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    n = int(rng.choice([300, 1500, 9000, 30000, 80000, 120000]))
    bsz = int(rng.integers(1, 4))
    b = synth_batch(max(n // bsz, 100), bsz, int(rng.integers(0, 1 << 30)))
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    loss = float(run(feats, coords, labels))
    assert np.isfinite(loss), (it, n, bsz, loss)
    mem = torch.cuda.max_memory_allocated() / 2**20
    if it % 5 == 0:
        print(f'it {it}: voxels {coords.shape[0]} batch {bsz} loss {loss:.4f} peak mem {mem:.0f} MiB', flush=True)
print(f'soak ok: {time.time() - t0:.1f} s, peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB')
