"""Follow-up of dbg_teacher_repro2.py (first deviation = the output of a decoder point_to_voxel): checksums of what the teacher's
point<->voxel transfers consume and produce -- index tensors, weights, counts, the CSR order, input and output rows -- per call,
compared with the first step's.   python tools/dbg_teacher_repro3.py [steps=60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from u2mkd_amd.torchsparse.nn import functional as F
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=(900, 1600)))
run = _runner(1.0, 2.0)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
log = []
main = torch.cuda.current_stream().cuda_stream


def cs(t):
    t = t.detach()
    if t.is_floating_point():
        return t.double().sum().reshape(1)
    w = torch.arange(1, t.numel() + 1, device=t.device, dtype=torch.float64) % 1000003
    return (t.reshape(-1).double() * w).sum().reshape(1)


def on_teacher():
    return torch.cuda.current_stream().cuda_stream != main


orig_vox, orig_devox, orig_seg = F.spvoxelize, F.spdevoxelize, F._segment_sum


def spvoxelize(feats, coords, counts):
    out = orig_vox(feats, coords, counts)
    if on_teacher() and feats.shape[1] >= 16:
        c32 = coords if coords.dtype == torch.int32 else coords.__dict__['_u2mkd_plans']['i32'][1]
        order, seg = c32.__dict__['_u2mkd_plans']['vox_csr_%d' % counts.shape[0]][1]
        live = order[:int(0) or None]
        log.append(('vox %dx%d->%d' % (feats.shape[0], feats.shape[1], counts.shape[0]),
                    torch.cat([cs(feats), cs(coords), cs(counts), cs(order), cs(seg), cs(out)])))
    return out


def spdevoxelize(feats, coords, weights):
    out = orig_devox(feats, coords, weights)
    if on_teacher() and feats.shape[1] >= 16:
        log.append(('devox %dx%d->%d' % (feats.shape[0], feats.shape[1], coords.shape[0]),
                    torch.cat([cs(feats), cs(coords), cs(weights), cs(out)])))
    return out


F.spvoxelize, F.spdevoxelize = spvoxelize, spdevoxelize
import u2mkd_amd.lidar.point_voxel as PV
PV.spf.spvoxelize, PV.spf.spdevoxelize = spvoxelize, spdevoxelize
ref = None
for i in range(steps):
    run.model.load_state_dict(state)
    log.clear()
    out, ld = _step(run, d, False)
    names = [n for n, _ in log]
    vals = [v.cpu() for _, v in log]
    if ref is None:
        ref = (names, vals)
        print(len(names), 'teacher transfers:', names, flush=True)
        continue
    assert names == ref[0], (names, ref[0])
    for n, v, r in zip(names, vals, ref[1]):
        if not torch.equal(v, r):
            print('step %d: %s differs in fields %s (vox: feats, idx, counts, order, seg, out; devox: feats, idx, weights, out)'
                  % (i, n, (v != r).nonzero().view(-1).tolist()), flush=True)
print('done', flush=True)
