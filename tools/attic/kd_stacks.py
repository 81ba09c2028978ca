"""Who issues the small torch kernels of a KD step: torch.profiler with Python stacks, grouped by (op, innermost
u2mkd_amd / torch.nn frame), for the ops that dominate the launch count."""
import os, sys, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from tools.kd_host import build
run, d = build(80000)
for _ in range(4): run(d)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, with_modules=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    run(d)
    torch.cuda.synchronize()
avg = prof.key_averages(group_by_stack_n=12)
rows = []
for e in avg:
    n_k = getattr(e, 'device_time_total', 0)
    if e.key in ('aten::copy_', 'aten::fill_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::zeros', 'aten::zero_', 'aten::cat',
                 'aten::where', 'aten::arange', 'aten::div', 'aten::sum', 'aten::clone', 'aten::contiguous', 'aten::to',
                 'aten::_to_copy', 'aten::index_select', 'aten::index', 'aten::full', 'aten::bitwise_or', 'aten::bitwise_and'):
        site = ''
        for fr in e.stack:
            if 'u2mkd_amd' in fr or 'torch/nn/modules' in fr or 'autograd' in fr:
                site = fr.split('/')[-1] if 'u2mkd_amd' not in fr else fr.split('u2mkd_amd/')[-1]
                if 'u2mkd_amd' in fr:
                    break
        rows.append((e.count, e.key, site[:90]))
rows.sort(reverse=True)
print('stacks available:', sum(1 for e in avg if e.stack))
for c, k, s in rows[:70]:
    print(f'{c:5d} {k:22s} {s}')
