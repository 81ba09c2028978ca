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
# the probe is not in the shipped library: tools/build_variant.sh _probe "-DU2MKD_DEBUG_PROBE" voxel.hip, then
# U2MKD_LIB_SUFFIX=_probe python tools/dbg_stale_probe.py
_i32, _i64, _p = C.c_int32, C.c_int64, C.c_void_p
L.SIGNATURES.update({'u2mkd_debug_probe_read': (C.c_int, [_p, _i64, _p, _i32]),
                     'u2mkd_debug_probe_entry_bytes': (_i32, []),
                     'u2mkd_debug_probe_rows_read': (C.c_int, [_p, _i32]),
                     'u2mkd_debug_probe_wg_read': (C.c_int, [_p, _p, _i32])})
lib = L.load()
assert lib.u2mkd_debug_probe_entry_bytes() == ENTRY.itemsize, (lib.u2mkd_debug_probe_entry_bytes(), ENTRY.itemsize)
CAP = 4096


def read_log():
    buf = np.zeros(CAP, ENTRY)
    n = C.c_int32(0)
    L.call('u2mkd_debug_probe_read', buf.ctypes.data, CAP, C.addressof(n), 1)
    return buf[:min(n.value, CAP)], n.value


WG_RING, WG_MAX = 64, 2048
calls = []            # this step's probe launches: (launch number, n, scale, w, i8, stream)
_launch = [0]


def ti_weights_sentinel(coords, idx_query_kn, scale=1):
    """functional.ti_weights_n8 with its two outputs pre-filled (NaN / -7777) on the launch stream: a row the launch did not
    write stays recognisable."""
    coords = coords.contiguous().float()
    idx = idx_query_kn.contiguous()
    n = coords.shape[0]
    w = torch.full((n, 8), float('nan'), dtype=torch.float32, device=coords.device)
    i8 = torch.full((n, 8), -7777, dtype=torch.int32, device=coords.device)
    L.call('u2mkd_ti_weights', L.ptr(coords), L.ptr(idx), n, float(scale), L.ptr(w), L.ptr(i8), L.stream())
    calls.append((_launch[0], n, scale, w, i8, torch.cuda.current_stream().cuda_stream))
    _launch[0] += 1
    return w, i8


from u2mkd_amd.torchsparse.nn import functional as _F
_F.ti_weights_n8 = ti_weights_sentinel


def read_wg():
    buf = np.zeros((WG_RING, WG_MAX, 2), np.uint32)
    n = C.c_int32(0)
    L.call('u2mkd_debug_probe_wg_read', buf.ctypes.data, C.addressof(n), 1)
    return buf, n.value


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (360, 640)
d = T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234, image_hw=hw))
run = _runner(1.0, 2.0)
state = {k: v.clone() for k, v in run.model.state_dict().items()}
prev, all_rec, dev_steps, rec_steps = None, [], [], []
unwritten_steps, ghost_steps, value_steps = [], [], []
ref_calls = None
main_stream = torch.cuda.current_stream().cuda_stream
read_wg()
for step in range(steps):
    run.model.load_state_dict(state)
    calls.clear()
    out, ld = _step(run, d, False)
    rec, n = read_log()
    wg, total = read_wg()
    assert total == _launch[0], (total, _launch[0])
    mine = set()
    for ln, npts, scale, w, i8, st in calls:
        nb = (npts + 255) // 256
        mine.add(ln + 1)
        row = wg[ln % WG_RING, :min(nb, WG_MAX)]
        missing = np.nonzero(row[:, 0] != ln + 1)[0]
        wrongptr = np.nonzero((row[:, 0] == ln + 1) & (row[:, 1] != (w.data_ptr() & 0xffffffff)))[0]
        bad_w = torch.isnan(w).any(1)
        bad_i = (i8 == -7777).any(1)
        nbw, nbi = int(bad_w.sum()), int(bad_i.sum())
        if len(missing) or len(wrongptr) or nbw or nbi:
            unwritten_steps.append(step)
            rows = bad_w.nonzero().view(-1)
            print('step %d launch %d (n %d scale %s stream %#x): %d workgroups without a record %s, %d with another output pointer; '
                  '%d weight rows / %d index rows NEVER WRITTEN (rows %s.., workgroups %s)'
                  % (step, ln, npts, scale, st, len(missing), missing[:8].tolist(), len(wrongptr), nbw, nbi, rows[:4].tolist(),
                     torch.unique(rows // 256)[:12].tolist()), flush=True)
    # values: every launch against the same launch of the first step (the batch and the state are the same every step)
    cur = [(npts, scale, st == main_stream, w.cpu(), i8.cpu()) for ln, npts, scale, w, i8, st in calls]
    if ref_calls is None:
        ref_calls = cur
        print('%d ti_weights launches per step: %s' % (len(cur), [('main' if m else 'side', n_, sc) for n_, sc, m, _, _ in cur]), flush=True)
    else:
        for j, ((n_, sc, m, w, i8), (_, _, _, rw, ri)) in enumerate(zip(cur, ref_calls)):
            dw = (w != rw).any(1)
            di = (i8 != ri).any(1)
            if bool(dw.any()) or bool(di.any()):
                value_steps.append(step)
                rows = dw.nonzero().view(-1)
                print('step %d: launch #%d of the step (%s stream, n %d, scale %s): %d weight rows / %d index rows differ from step 0; rows %s .. %s, workgroups %s'
                      % (step, j, 'main' if m else 'side', n_, sc, int(dw.sum()), int(di.sum()), rows[:4].tolist(), rows[-2:].tolist(),
                         torch.unique(rows // 256)[:16].tolist()), flush=True)
                dbg2 = np.zeros((2, 81920, 4), np.uint32)
                L.call('u2mkd_debug_probe_rows_read', dbg2.ctypes.data, calls[j][0] % 16)
                dbg, tim = dbg2[0], dbg2[1]
                bad = rows.numpy()
                good = np.setdiff1d(np.arange(n_), bad)
                print('     thread durations (10 ns ticks): deviating rows median %d max %d; all other rows median %d, 99.9th percentile %d, max %d; TRAPSTS of deviating rows %s, of the others %s; STATUS %s / %s'
                      % (np.median(tim[bad, 0]), tim[bad, 0].max(), np.median(tim[good, 0]), np.percentile(tim[good, 0], 99.9), tim[good, 0].max(),
                         [hex(v) for v in np.unique(tim[bad, 1])[:6]], [hex(v) for v in np.unique(tim[good, 1])[:6]],
                         [hex(v) for v in np.unique(tim[bad, 2])[:6]], [hex(v) for v in np.unique(tim[good, 2])[:6]]), flush=True)
                for r in rows[:4].tolist():
                    miss = int(dbg[r, 0]) & 0xff
                    print('     row %d now %s | step 0 %s | idx now %s | the thread SAW idx == -1 at corners %s (xcc %d), computed w[0] = %.4f, fractions %s, launch %d (expected %d)'
                          % (r, [round(x, 4) for x in w[r].tolist()], [round(x, 4) for x in rw[r].tolist()], i8[r].tolist(),
                             [k for k in range(8) if miss >> k & 1], int(dbg[r, 0]) >> 8 & 0xf, float(dbg[r, 1:2].view(np.float32)[0]),
                             [int(dbg[r, 2]) & 0xff, int(dbg[r, 2]) >> 8 & 0xff, int(dbg[r, 2]) >> 16 & 0xff], int(dbg[r, 3]), calls[j][0]), flush=True)
    ghosts = [(r, b, int(wg[r, b, 0]) - 1, int(wg[r, b, 1])) for r, b in zip(*np.nonzero((wg[:, :, 0] != 0) & ~np.isin(wg[:, :, 0], list(mine))))]
    if ghosts:
        ghost_steps.append(step)
        print('step %d: %d workgroup records that belong to NO launch of this step (ran on old arguments): first %s; this step\'s launches %s'
              % (step, len(ghosts), ghosts[:6], sorted(mine)[:3]), flush=True)
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
print('steps %d: %d with rows never written / workgroups without a record %s; %d with ghost records %s' % (steps, len(unwritten_steps), unwritten_steps[:20], len(ghost_steps), ghost_steps[:20]))
print('steps %d: %d with ti_weights outputs that differ from step 0: %s' % (steps, len(set(value_steps)), sorted(set(value_steps))[:30]))
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
