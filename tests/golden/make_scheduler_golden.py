"""LR-schedule fixture: values of the REFERENCE's cosine_schedule_with_warmup (core/schedulers.py:10-35) at a
grid of (step, world size, epochs, batch, dataset size), produced by importing the reference file in the build
container with torchpack.distributed.size() stubbed.  `python tests/golden/make_scheduler_golden.py`."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
    tp = types.ModuleType('torchpack')
    td = types.ModuleType('torchpack.distributed')
    world = {'n': 1}
    td.size = lambda: world['n']
    sys.modules.update({'torchpack': tp, 'torchpack.distributed': td})
    sys.path.insert(0, REF)
    src = open(os.path.join(REF, 'core', 'schedulers.py')).read()
    # the file's tail (PolyLR) needs a `configs` global that only exists under the full trainer: the function under
    # test is self-contained
    ns = {}
    exec(compile(src.split('class LambdaStepLR')[0], 'schedulers.py', 'exec'), ns)
    fn = ns['cosine_schedule_with_warmup']
    rows = []
    for w in (1, 2, 4, 8):
        world['n'] = w
        for epochs, bs, size in ((25, 2, 28130), (50, 3, 28130), (1, 1, 100)):
            for k in (0, 1, 5, 124, 125, 126, 249, 250, 499, 500, 999, 1000, 5000, 14065, 100000, 351625):
                rows.append({'k': k, 'world': w, 'num_epochs': epochs, 'batch_size': bs, 'dataset_size': size,
                             'value': float(fn(k, epochs, bs, size))})
    with open(os.path.join(HERE, 'cosine_schedule.json'), 'w') as f:
        json.dump(rows, f)
    print(len(rows), 'rows')


if __name__ == '__main__':
    main()
