"""Is the teacher's trilinear-weight tensor overwritten after it was computed, or computed from different inputs?  Every
ti_weights_n8 call on a side stream keeps a clone of (coords, idx_kn, weights) made right behind the kernel; at the end of the
step the live tensors are compared with the clones (overwritten later?) and the clones with the first step's (inputs differ?).
    python tools/dbg_teacher_repro4.py [steps=60] [H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from u2mkd_amd.torchsparse.nn import functional as F
import u2mkd_amd.lidar.point_voxel as PV
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (360, 640)
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=hw))
run = _runner(1.0, 2.0)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
main = torch.cuda.current_stream().cuda_stream
calls = []
orig = F.ti_weights_n8


def ti(coords, idx_kn, scale=1):
    w, i8 = orig(coords, idx_kn, scale)
    if torch.cuda.current_stream().cuda_stream != main:
        calls.append((scale, coords, idx_kn, w, i8, coords.clone(), idx_kn.clone(), w.clone(), i8.clone()))
    return w, i8


F.ti_weights_n8 = ti
PV.spf.ti_weights_n8 = ti
ref = None
for step in range(steps):
    run.model.load_state_dict(state)
    calls.clear()
    out, ld = _step(run, d, False)
    torch.cuda.synchronize()
    cur = []
    for scale, c, k, w, i8, c0, k0, w0, i0 in calls:
        late = [n for n, a, b in (('coords', c, c0), ('idx_kn', k, k0), ('weights', w, w0), ('idx8', i8, i0)) if not torch.equal(a, b)]
        if late:
            bad = (w != w0).any(1).nonzero().view(-1) if 'weights' in late else torch.empty(0)
            print('step %d scale %s: OVERWRITTEN after the kernel: %s (%d weight rows, first %s)'
                  % (step, scale, late, bad.numel(), bad[:5].tolist()), flush=True)
        cur.append((scale, c0.cpu(), k0.cpu(), w0.cpu()))
    if ref is None:
        ref = cur
        print('%d side-stream ti_weights calls per step, scales %s' % (len(cur), [s for s, *_ in cur]), flush=True)
        continue
    for (s, c0, k0, w0), (_, rc, rk, rw) in zip(cur, ref):
        diff = [n for n, a, b in (('coords', c0, rc), ('idx_kn', k0, rk), ('weights', w0, rw)) if not torch.equal(a, b)]
        if diff:
            print('step %d scale %s: differs from the first step AT THE KERNEL: %s' % (step, s, diff), flush=True)
            if 'weights' in diff:
                rows = (w0 != rw).any(1).nonzero().view(-1)
                print('   %d weight rows differ; rows %s .. %s' % (rows.numel(), rows[:4].tolist(), rows[-2:].tolist()), flush=True)
                for r in rows[:3].tolist():
                    print('   row %d: now %s\n            ref %s\n            idx_kn %s coords %s' % (r, [round(v, 5) for v in w0[r].tolist()], [round(v, 5) for v in rw[r].tolist()],
                          k0[:, r].tolist(), c0[r].tolist()), flush=True)
print('done', flush=True)
