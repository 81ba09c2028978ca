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

# the KD step (five streams, deferred weight gradients, geometry pre-pass): losses finite over changing scene sizes, memory bounded
from u2mkd_amd import kd as KD
from u2mkd_amd.synth import synth_kd_batch
torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
kd_model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
kd_run = T.KDStep(kd_model, num_epochs=50, batch_size=1)
kd_run.train_mode()
t0 = time.time()
torch.cuda.reset_peak_memory_stats()
sizes = [int(rng.choice([2000, 9000, 30000, 80000])) for _ in range((int(sys.argv[2]) if len(sys.argv) > 2 else 30) + 1)]
batches = [T.kd_batch_to_device(synth_kd_batch(n, 1, seed=100 + i, image_hw=(360, 640))) for i, n in enumerate(sizes[:6])]
cur = dict(T.fresh_batch(batches[0]), _key=0)
watch = T.TeacherWatch(kd_model.model_t)     # the frozen teacher must give bit-identical logits whenever a batch comes round again
for it in range(len(sizes) - 1):
    nxt = dict(T.fresh_batch(batches[(it + 1) % len(batches)]), _key=(it + 1) % len(batches))      # (the batch carries its key: its teacher may run a step ahead)
    watch.key = it % len(batches)
    loss = float(kd_run(cur, prefetch=nxt))
    assert np.isfinite(loss), (it, loss)
    cur = nxt
    if it % 5 == 0:
        print(f'kd it {it}: points {sizes[it % len(batches)]} loss {loss:.4f} peak mem {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB', flush=True)
bad, compared = watch.deviating_steps()
assert bad == 0, f'the frozen teacher deviated in {bad} of {compared} revisited steps'
print(f'kd soak: teacher bit-identical in all {compared} revisited steps')
print(f'kd soak ok: {time.time() - t0:.1f} s, peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB')
