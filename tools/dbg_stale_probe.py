"""In-kernel evidence for (or against) stale reads behind a same-stream producer.

u2mkd_ti_weights runs its probe variant (csrc/voxel.hip, U2MKD_DEBUG_TI_PROBE=1): every input word is read with an ordinary
load, with an agent-scope load, with a system-scope load and once more with an ordinary load behind `buffer_inv sc0 sc1`; a
thread whose reads disagree logs (point, word, the four values, XCC id, hardware id, wall clock).  The inputs were written by
kernels that precede the probe on the SAME stream (hash query -> idx_kn; element-wise -> coords) and nothing writes them while
the probe runs, so any record is a read that did not observe its in-order producer.

    python tools/dbg_stale_probe.py [steps=120] [H W]

Prints, per step: records, the teacher rows that differ from the previous step; at the end a summary by XCC / value pattern."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['U2MKD_DEBUG_TI_PROBE'] = '1'
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import numpy as np
import torch
from u2mkd_amd import _lib as L, train as T
from u2mkd_amd.synth import synth_kd_batch
from test_gpu_configs import _runner
from test_gpu_configs4_fullsize import _step

ENTRY = np.dtype([('i', '<i8'), ('k', '<i4'), ('xcc', '<u4'), ('hwid', '<u4'), ('launch', '<u4'), ('v_plain', '<i8'),
                  ('v_agent', '<i8'), ('v_sys', '<i8'), ('v_after_inv', '<i8'), ('t', '<u8')])
lib = L.load()
assert lib.u2mkd_debug_probe_entry_bytes() == ENTRY.itemsize, (lib.u2mkd_debug_probe_entry_bytes(), ENTRY.itemsize)
CAP = 4096


def read_log():
    buf = np.zeros(CAP, ENTRY)
    n = C.c_int32(0)
    L.call('u2mkd_debug_probe_read', buf.ctypes.data, CAP, C.addressof(n), 1)
    return buf[:min(n.value, CAP)], n.value


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (360, 640)
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=hw))
run = _runner(1.0, 2.0)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
prev, all_rec, dev_steps, rec_steps = None, [], [], []
for step in range(steps):
    run.model.load_state_dict(state)
    out, ld = _step(run, d, False)
    rec, n = read_log()
    t = out['t']['x_vox'].clone()
    rows = 0
    if prev is not None:
        dt = (t - prev).abs()
        rows = int((dt.max(1).values > 0).sum())
    if rows:
        dev_steps.append(step)
    if n:
        rec_steps.append(step)
        all_rec.append(rec)
    if rows or n:
        print('step %d: %d probe records, %d teacher rows differ from the previous step' % (step, n, rows), flush=True)
        for e in rec[:6]:
            print('   point %d word %d launch %d xcc %#x hwid %#x: plain %d agent %d system %d plain-after-inv %d t %d'
                  % (e['i'], e['k'], e['launch'], e['xcc'], e['hwid'], e['v_plain'], e['v_agent'], e['v_sys'], e['v_after_inv'], e['t']), flush=True)
    prev = t
print('steps %d: %d with probe records %s; %d with deviating teacher rows %s' % (steps, len(rec_steps), rec_steps[:20], len(dev_steps), dev_steps[:20]))
if all_rec:
    r = np.concatenate(all_rec)
    print('records %d; words: idx %d, coords %d' % (len(r), int((r['k'] < 8).sum()), int((r['k'] >= 8).sum())))
    print('  plain != system: %d; agent != system: %d; plain-after-invalidate != system: %d'
          % (int((r['v_plain'] != r['v_sys']).sum()), int((r['v_agent'] != r['v_sys']).sum()), int((r['v_after_inv'] != r['v_sys']).sum())))
    print('  stale plain value == -1: %d' % int(((r['v_plain'] != r['v_sys']) & (r['v_plain'] == -1)).sum()))
    xs, cnt = np.unique(r['xcc'] & 0xf, return_counts=True)
    print('  by reading XCC:', dict(zip(xs.tolist(), cnt.tolist())))
    for rec in all_rec[:4]:
        pts = np.unique(rec['i'])
        print('  one step: %d records over %d points, point range %d..%d, 256-point workgroups %s, time span %d ticks'
              % (len(rec), len(pts), pts.min(), pts.max(), np.unique(pts // 256)[:12].tolist(), int(rec['t'].max() - rec['t'].min())))
print('done', flush=True)
