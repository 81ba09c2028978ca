"""Launch census of one KD training step: device kernels grouped by the u2mkd_amd source line that issued them
(torch.profiler, forward ops by Python stack; backward kernels are attributed to their autograd node)."""
import sys, collections; sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
from tools.kd_host import build
run, d = build(80000)
from u2mkd_amd import train as T
for _ in range(4): run(T.fresh_batch(d))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run(T.fresh_batch(d))
    torch.cuda.synchronize()
ev = prof.events()
# map: every CPU op with device kernels -> (count, device time), keyed by innermost repo frame + op name
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type.name != 'CPU' or not e.kernels:
        continue
    if e.cpu_parent is not None and e.cpu_parent.kernels:
        continue          # count kernels once, at the outermost op that owns them
    site = 'autograd/other'
    for fr in (e.stack or []):
        if 'u2mkd_amd' in fr:
            site = fr.split('u2mkd_amd/')[-1].strip()
            break
    k = (site[:70], e.name[:40])
    agg[k][0] += len(e.kernels)
    agg[k][1] += sum(kk.duration for kk in e.kernels)
tot_n = sum(v[0] for v in agg.values()); tot_t = sum(v[1] for v in agg.values())
print(f'kernels {tot_n}, device time {tot_t/1e3:.1f} ms')
bysite = collections.defaultdict(lambda: [0, 0.0])
for (site, name), v in agg.items():
    bysite[site][0] += v[0]; bysite[site][1] += v[1]
print('--- by source line (top 70 by launches)')
for site, v in sorted(bysite.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f'{v[0]:5d} {v[1]/1e3:8.2f} ms  {site}')
print('--- by (site, op) (top 50 by device time)')
for (site, name), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f'{v[0]:5d} {v[1]/1e3:8.2f} ms  {site} | {name}')
