"""Host floor of the SPVCNN training step: the same step on a tiny scene (GPU work ~0), so the
wall time per step is the Python + launch overhead the full-size step has to hide."""
import sys, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import lidar, train as T
from u2mkd_amd.synth import synth_batch
for n in (2000, 80000):
    b = synth_batch(n, 1)
    feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
    model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    run = T.LidarStep(model)
    for _ in range(5): run(feats, coords, labels)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): run(feats, coords, labels)
    torch.cuda.synchronize()
    print(f'N={n}: {(time.perf_counter() - t) / 10 * 1e3:.1f} ms/step', flush=True)
    if n == 2000:
        # forward / backward split of the host time
        tf = tb = 0.0
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = run.forward_loss(feats, coords, labels) if hasattr(run, 'forward_loss') else None
            if loss is None: break
            torch.cuda.synchronize(); t1 = time.perf_counter()
            loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
            tf += t1 - t0; tb += t2 - t1
        if tf: print(f'   fwd {tf*100:.1f} ms  bwd {tb*100:.1f} ms')
