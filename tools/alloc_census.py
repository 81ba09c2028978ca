"""Allocations (aten::empty / empty_like / empty_strided / new_empty...) of one KD step by the u2mkd_amd source line that made them.
python tools/alloc_census.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(4):
    step()
count = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if 'empty' in name or name in ('aten::zeros', 'aten::new_zeros', 'aten::full', 'aten::zeros_like', 'aten::clone', 'aten::_to_copy'):
            site = '?'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if '/u2mkd_amd/' in fr.filename:
                    site = '%s:%d %s' % (fr.filename.split('/u2mkd_amd/')[-1], fr.lineno, fr.name)
                    break
            count[(site, name)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
torch.cuda.synchronize()
print('allocating ops of one step: %d' % sum(count.values()))
for (site, name), v in count.most_common(60):
    print('  %4d  %-70s %s' % (v, site, name))
