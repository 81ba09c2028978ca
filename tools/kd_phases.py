"""Host-side timeline of the KD step: when does the Python thread leave each phase (launch issue incl. the waits inside),
and how long after the last phase has the GPU drained?  python tools/kd_phases.py [steps]"""
import sys, time; sys.path.insert(0, '.')
import torch
from u2mkd_amd import kd as KD, torchsparse as ts, train as T
from tools.kd_host import build

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
run, d = build(80000)
marks = []
orig_t = run.model.model_t.forward
orig_s = run.model.model_s.forward
orig_head = run.model.model_s.camera_head


def timed(name, fn):
    def f(*a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        marks[-1][name] = marks[-1].get(name, 0.0) + time.perf_counter() - t0
        return out
    return f


run.model.model_t.forward = timed('teacher_fwd', orig_t)
run.model.model_s.forward = timed('student_fwd', orig_s)
run.model.model_s.camera_head = timed('camera_head', orig_head)


def step(dd):
    m = {}
    marks.append(m)
    t0 = time.perf_counter()
    dd = T.fresh_batch(dd)
    m['clone'] = time.perf_counter() - t0
    stu = {'lidar': ts.SparseTensor(dd['s_feats'], dd['s_coords']), 'images': dd['images'],
           'pixel_coordinates': dd['pixel_coordinates'], 'masks': dd['masks'], 'fov_mask': dd['fov_mask']}
    tea = {'lidar': ts.SparseTensor(dd['t_feats'], dd['t_coords'])}
    t1 = time.perf_counter()
    out = run.net({'student': stu, 'teacher': tea})
    t2 = time.perf_counter()
    ld = KD.kd_losses(out, dd['targets'], dd['fov_mask'], dd['inverse_map'], dd['inds'], dd['num_pts'], dd['num_vox_t'],
                      run.crit, dd['keyframe_mask_full'])
    t3 = time.perf_counter()
    run.opt.zero_grad()
    ld['total'].backward()
    t4 = time.perf_counter()
    run.opt.step()
    run.sched.step()
    t5 = time.perf_counter()
    m.update(forward=t2 - t1, losses=t3 - t2, backward=t4 - t3, optimizer=t5 - t4, host_total=t5 - t0)


for _ in range(5):
    step(d)
torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
for _ in range(steps):
    step(d)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
keys = ['clone', 'camera_head', 'teacher_fwd', 'student_fwd', 'forward', 'losses', 'backward', 'optimizer', 'host_total']
print('per step (ms): ' + '  '.join('%s %.1f' % (k, 1e3 * sum(m.get(k, 0) for m in marks) / steps) for k in keys))
print('wall %.1f ms/step; host returned after %.1f ms/step; final drain %.1f ms' % (1e3 * wall / steps, 1e3 * t_host / steps, 1e3 * (wall - t_host)))
