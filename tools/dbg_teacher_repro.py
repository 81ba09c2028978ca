"""How often, and by how much, the frozen teacher's logits differ between two KD steps from the same state on the same batch
(the teacher's kernels are order-deterministic; DESIGN / NOTES N6: a rare last-place difference under stream concurrency that
a hard quantiser can amplify).   python tools/dbg_teacher_repro.py [steps=12] [H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
from u2mkd_amd import train as T
from u2mkd_amd.synth import synth_kd_batch
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

if os.environ.get('DBG_FENCE') == '1':        # an event record (a barrier packet with a system-scope release) behind every hash query
    from u2mkd_amd.torchsparse.nn import functional as _F
    _q = _F.HashTable.query

    def _query(self, queries):
        out = _q(self, queries)
        torch.cuda.current_stream().record_event()
        return out
    _F.HashTable.query = _query
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (900, 1600)
pts, sweeps = int(os.environ.get('DBG_POINTS', '80000')), int(os.environ.get('DBG_SWEEPS', '0')) or None      # (300000 / 9: configs[4]'s teacher scene)
d = T.kd_batch_to_device(synth_kd_batch(pts, 1, seed=1234, image_hw=hw, sweeps=sweeps))
amp = os.environ.get('DBG_AMP') or False          # DBG_AMP=bf16: the step under bf16 autocast (bf16 rows, library kernels in bf16)
run = _runner(float(os.environ.get('DBG_CR', '1.0')), 2.0, amp=amp)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
prev, rows_total = None, 0
for i in range(steps):
    run.model.load_state_dict(state)
    out, ld = _step(run, d, bool(amp))
    t = out['t']['x_vox'].clone()
    if prev is not None:
        dt = (t - prev).abs()
        rows = int((dt.max(1).values > 0).sum())
        rows_total += int(rows > 0)
        print('step %d: %d of %d teacher rows differ, max |diff| %.3g (max |logit| %.3g)' % (i, rows, t.shape[0], float(dt.max()), float(t.abs().max())), flush=True)
    prev = t
print('SUMMARY: %d steps, %d deviating comparisons' % (steps, rows_total), flush=True)
