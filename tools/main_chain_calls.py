"""What the MAIN stream of one pipelined KD step is asked to run that is not arithmetic of the step: every C-ABI call and every
sorting / scanning / indexing aten op issued while torch's current stream is the step's own stream, by the u2mkd_amd source line
(two frames) that issued it.  One steady-state step.
  python tools/main_chain_calls.py [pattern ...]      (default patterns: the plan / geometry-like entries)"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from u2mkd_amd import _lib as L

pats = sys.argv[1:] or ['csr', 'kmap', 'table', 'hash', 'schedule', 'sort', 'plan', 'keys', 'ranges', 'quant', 'l2c_finish', 'unique', 'cumsum',
                        'scan', 'mailbox', 'count', 'index_select', 'nonzero', 'bucket', 'searchsorted']
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(6):
    step()
torch.cuda.synchronize()
main = torch.cuda.current_stream().cuda_stream
cnt = collections.Counter()
on = [False]


def site():
    fr = [f for f in traceback.extract_stack(limit=30) if '/u2mkd_amd/' in f.filename and not f.filename.endswith('_lib.py')]
    fr = fr[-3:]
    return ' < '.join('%s:%d %s' % (f.filename.split('/u2mkd_amd/')[-1], f.lineno, f.name) for f in reversed(fr))


real_call = L.call


def call(name, *a):
    if on[0] and torch.cuda.current_stream().cuda_stream == main and any(p in name for p in pats):
        cnt[(name, site())] += 1
    return real_call(name, *a)


L.call = call
import u2mkd_amd.torchsparse.nn.functional as spf
import u2mkd_amd.sptr.functional as sf
import u2mkd_amd.fusion as fu
import u2mkd_amd.pixel_head as ph
for m in (spf, sf, fu, ph):
    if getattr(m, 'L', None) is L:
        pass      # (modules hold the module object: the patched attribute is seen)


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if on[0] and torch.cuda.current_stream().cuda_stream == main and any(p in name for p in pats):
            cnt[(name, site())] += 1
        return out


on[0] = True
with Mode():
    step()
on[0] = False
torch.cuda.synchronize()
for (name, s), c in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print('%3d  %-34s %s' % (c, name, s))
