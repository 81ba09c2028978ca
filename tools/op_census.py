"""Where the torch glue of one KD training step comes from: every aten op that launches device work, counted by the
u2mkd_amd source line that issued it.  Forward ops by their Python stack; backward ops by the forward stack of the
autograd node that runs them (anomaly mode keeps it) -- the hand-written HIP operators show up as their autograd
Function and are listed for scale only.  One step, after warm-up.
usage: python tools/op_census.py [n_voxels=80000]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import u2mkd_amd._lib as L
from tools.kd_host import build  # noqa: E402  (prints its own two lines)
from u2mkd_amd import train as T

NOLAUNCH = ('aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::t', 'aten::transpose', 'aten::permute', 'aten::expand',
            'aten::slice', 'aten::select', 'aten::unsqueeze', 'aten::squeeze', 'aten::detach', 'aten::alias', 'aten::as_strided',
            'aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::new_empty', 'aten::narrow', 'aten::unbind', 'aten::split',
            'aten::_local_scalar_dense', 'aten::lift_fresh', 'aten::is_same_size', 'aten::sym_size', 'aten::size', 'aten::stride',
            'aten::unfold', 'aten::view_as_real', 'aten::chunk', 'aten::flatten', 'aten::_reshape_alias', 'aten::is_pinned',
            'aten::record_stream', 'aten::set_', 'aten::resize_', 'aten::is_nonzero', 'aten::new_empty_strided', 'aten::split_with_sizes',
            'aten::_to_copy.NOCUDA')


def site_of(frames):
    for fr in reversed(frames):
        if '/u2mkd_amd/' in fr.filename and '/tools/' not in fr.filename:
            return '%s:%d %s' % (fr.filename.split('/u2mkd_amd/')[-1], fr.lineno, fr.name)
    return '?'


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.fwd = collections.Counter()
        self.bwd = collections.Counter()
        self.fwd_b = collections.Counter()       # bytes touched (inputs + outputs on the device): a proxy for the kernel time
        self.bwd_b = collections.Counter()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name in NOLAUNCH:
            return out
        flat = [a for a in (list(args) + list((kwargs or {}).values()) + (list(out) if isinstance(out, (tuple, list)) else [out])) if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in flat):
            return out
        nbytes = sum(t.numel() * t.element_size() for t in flat if t.is_cuda)
        node = torch._C._current_autograd_node()
        if node is None:
            key = (site_of(traceback.extract_stack(limit=40)), name)
            self.fwd[key] += 1
            self.fwd_b[key] += nbytes
        else:
            tb = node.metadata.get('traceback_', None)
            where = '?'
            if tb:
                for line in reversed(tb):
                    if '/u2mkd_amd/' in line:
                        f = line.strip().split('\n')[0]
                        where = f.split('/u2mkd_amd/')[-1].replace('", line ', ':').replace(', in ', ' ')
                        break
            key = (where, node.name().split('::')[-1][:36], name)
            self.bwd[key] += 1
            self.bwd_b[key] += nbytes
        return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
run, d = build(n)
for _ in range(3):
    run(T.fresh_batch(d))
torch.cuda.synchronize()
hip = collections.Counter()
real = L.call


def counting(name, *a):
    hip[name] += 1
    return real(name, *a)


L.call = counting
with torch.autograd.set_detect_anomaly(True, check_nan=False):
    c = Census()
    with c:
        run(T.fresh_batch(d))
torch.cuda.synchronize()
L.call = real
print('aten ops with device work: forward %d, backward %d; C-ABI calls %d' % (sum(c.fwd.values()), sum(c.bwd.values()), sum(hip.values())))
print('--- forward, by (site, op)')
for (s, o), v in sorted(c.fwd.items(), key=lambda kv: -kv[1])[:120]:
    print('%5d  %-70s %s' % (v, s, o))
print('--- backward, by (forward site of the node, node, op)')
for (s, nd, o), v in sorted(c.bwd.items(), key=lambda kv: -kv[1])[:120]:
    print('%5d  %-60s %-36s %s' % (v, s, nd, o))
print('--- by bytes touched (MB per step): forward')
for (s, o), v in sorted(c.fwd_b.items(), key=lambda kv: -kv[1])[:45]:
    print('%8.1f MB %4d  %-66s %s' % (v / 1e6, c.fwd[(s, o)], s, o))
print('--- by bytes touched (MB per step): backward')
for (s, nd, o), v in sorted(c.bwd_b.items(), key=lambda kv: -kv[1])[:45]:
    print('%8.1f MB %4d  %-56s %-34s %s' % (v / 1e6, c.bwd[(s, nd, o)], s, nd, o))
print('--- C-ABI entry points')
for k, v in hip.most_common(60):
    print('%5d  %s' % (v, k))
