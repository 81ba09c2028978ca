"""The autograd graph of one KD step by node type (what the engine walks in ``loss.backward()``: its own share of the host's
backward issue is ~20 ms).   python tools/autograd_nodes.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from u2mkd_amd import kd as KD

args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
run = step.runner
for _ in range(3):
    step()
seen = {}
real = run.amp.backward_and_step


def spy(loss, opt):
    stack, count = [loss.grad_fn], collections.Counter()
    visited = set()
    while stack:
        f = stack.pop()
        if f is None or f in visited:
            continue
        visited.add(f)
        count[type(f).__name__] += 1
        for g, _ in f.next_functions:
            stack.append(g)
    seen['nodes'] = count
    return real(loss, opt)


run.amp.backward_and_step = spy
step()
torch.cuda.synchronize()
c = seen['nodes']
print('autograd nodes of one KD step: %d' % sum(c.values()))
for k, v in c.most_common(45):
    print('  %5d  %s' % (v, k))
