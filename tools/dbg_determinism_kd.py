"""Teacher / student logits of two identical KD forwards (+backward) from one state: equal?  env knobs apply.
python tools/dbg_determinism_kd.py [n_pts] [sweeps] [bf16|f32] [reload]"""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
sw = int(sys.argv[2]) if len(sys.argv) > 2 else 9
amp = (sys.argv[3] if len(sys.argv) > 3 else 'bf16') == 'bf16'
reload_ = len(sys.argv) > 4 and sys.argv[4] == 'reload'
nb = synth_kd_batch(n, 1, seed=1234, image_hw=(64, 112), sweeps=sw)
d = T.kd_batch_to_device(nb)
run = _runner(2.0, 2.0, amp='bf16' if amp else False)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
res = []
for r in range(5):
    if reload_:
        run.model.load_state_dict(state)
    out, ld = _step(run, d, amp)
    res.append((out['t']['x_vox'].float().clone(), out['stu']['x_vox'].detach().float().clone(), float(ld['total'])))
for r in (1, 2, 3, 4):
    dt = (res[r][0] - res[r - 1][0]).abs()
    ds = (res[r][1] - res[r - 1][1]).abs()
    print('run', r, 'vs', r - 1, 'teacher equal', torch.equal(res[r][0], res[r - 1][0]), 'max diff %.3g rows differing %d' % (float(dt.max()), int((dt.max(1).values > 0).sum())),
          '| student equal', torch.equal(res[r][1], res[r - 1][1]), 'max diff %.3g' % float(ds.max()), '| loss', res[r][2], res[r - 1][2])
