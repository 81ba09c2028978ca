"""gpurun_out/{prof_<tag>, prof_<tag>_ko, pmc_<tag>_*} -> profiles/<round>_*: rocprofv3 kernel
stats of the default bench and of `bench.py --kernel-only`, the PMC means per kernel, and the HBM
traffic per launch of the roofline group (FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B
on this counter set; FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md prescribes).
usage: python tools/summarize_profiles.py <tag> <round>"""
import csv, json, os, shutil, sys, collections
tag, rnd = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
shutil.copy(os.path.join(G, f'prof_{tag}', 'bench_kernel_stats.csv'), os.path.join(P, f'{rnd}_default_kernel_stats.csv'))
shutil.copy(os.path.join(G, f'prof_{tag}_ko', 'bench_kernel_stats.csv'), os.path.join(P, f'{rnd}_kernel_only_kernel_stats.csv'))
line = open(os.path.join(G, f'prof_{tag}', 'bench_stdout.txt')).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(P, f'{rnd}_bench_under_rocprof.json'), 'w').write(line + '\n')
acc = collections.defaultdict(lambda: [0.0, 0])
for sub in ('sq', 'sq2', 'fetch', 'write'):
    f = os.path.join(G, f'pmc_{tag}_{sub}', 'pmc_counter_collection.csv')
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        a = acc[(r['Kernel_Name'], r['Counter_Name'])]
        a[0] += float(r['Counter_Value']); a[1] += 1
with open(os.path.join(P, f'{rnd}_kernel_only_pmc_summary.csv'), 'w') as f:
    f.write('kernel,counter,dispatches,mean\n')
    for (k, c), (s, n) in acc.items():
        kk = k.split('(')[0]
        f.write(f'"{kk}",{c},{n},{s / n}\n')
def mean(kname, counter, exact=True):
    """mean of a counter over the dispatches of ONE kernel: `kname` is the full name up to its argument list (template
    arguments included -- the fp32 and the bf16-storage instantiations of conv_tp / conv_wgrad_x3 move different bytes;
    rounds 1-3 matched by substring and averaged them)"""
    tot = [v for (k, c), v in acc.items() if c == counter and (k.split('(')[0] == kname if exact else kname in k)]
    return sum(s for s, n in tot) / max(sum(n for s, n in tot), 1)
def traffic(kname):   # bytes per launch
    return (2.0 * mean(kname, 'FETCH_SIZE') + mean(kname, 'WRITE_SIZE')) * 1024.0
names = sorted({k.split('(')[0] for (k, c) in acc})
def full(prefix, *templ):
    """the profiled kernel whose name starts with `prefix` and carries the template arguments `templ`"""
    hits = [k for k in names if k.startswith(prefix) and all(t in k for t in templ)]
    assert len(hits) == 1, (prefix, templ, hits)
    return hits[0]
# the fp32-row tile kernel of the run: f16x2 (AR = 4, the default) or bf16x3 (AR = 2, U2MKD_CONV_ARITH=bf16x3)
K_TP32 = (full('void u2mkd::conv_tp_kernel<4, 1, 64, false, 4') if any(k.startswith('void u2mkd::conv_tp_kernel<4, 1, 64, false, 4') for k in names)
          else full('void u2mkd::conv_tp_kernel<4, 1, 64, false, 2'))
K_TP16 = full('void u2mkd::conv_tp_kernel<4, 1, 64, false, 3')
K_WG32, K_WG16 = full('void u2mkd::conv_wgrad_x3_kernel<false>'), full('void u2mkd::conv_wgrad_x3_kernel<true>')
K_RED, K_FRAG = full('u2mkd::wgrad_pairs_reduce_kernel'), full('u2mkd::weight_fragments_batch_kernel')
meta = json.loads(line)['roofline']
nfrag = meta['fragments']['weights_per_launch']
t = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --kernel-only`; '
               'bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of MI355X_MICROARCH.md); '
               'kernels matched by their full template name',
     'N': meta['N'], 'P': meta['P'],
     'kernels': {'conv_tp': K_TP32, 'conv_wgrad': K_WG32, 'wgrad_reduce': K_RED, 'weight_fragments_batch': K_FRAG},
     'conv_tp_fwd_or_dgrad': traffic(K_TP32),
     'weight_fragments_share': traffic(K_FRAG) / nfrag,
     'conv_wgrad_pairs': traffic(K_WG32),
     'wgrad_reduce': traffic(K_RED)}
t['group_fwd_dgrad_wgrad'] = 2 * t['conv_tp_fwd_or_dgrad'] + t['weight_fragments_share'] + t['conv_wgrad_pairs'] + t['wgrad_reduce']
t['bf16_storage'] = {'kernels': {'conv_tp': K_TP16, 'conv_wgrad': K_WG16, 'wgrad_reduce': K_RED},
                     'conv_tp_fwd_or_dgrad': traffic(K_TP16), 'conv_wgrad_pairs': traffic(K_WG16), 'wgrad_reduce': traffic(K_RED)}
t['bf16_storage']['group_fwd_dgrad_wgrad'] = (2 * t['bf16_storage']['conv_tp_fwd_or_dgrad'] + t['bf16_storage']['conv_wgrad_pairs']
                                               + t['bf16_storage']['wgrad_reduce'])
t['algorithmic_bytes'] = meta['algorithmic_bytes']
t['traffic_over_algorithmic'] = t['group_fwd_dgrad_wgrad'] / meta['algorithmic_bytes']
json.dump(t, open(os.path.join(P, f'{rnd}_traffic.json'), 'w'), indent=1)
print(json.dumps(t, indent=1))
for k in (K_TP32, K_WG32, K_TP16, K_WG16):
    busy, wave, mf = mean(k, 'SQ_BUSY_CYCLES'), mean(k, 'SQ_WAVE_CYCLES'), mean(k, 'SQ_VALU_MFMA_BUSY_CYCLES')
    print(k, {c: round(mean(k, c)) for c in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD')})
    print(k, 'MFMA_BUSY/WAVE_CYCLES', mf / max(wave, 1), 'WAIT_ANY/WAVE', mean(k, 'SQ_WAIT_ANY') / max(wave, 1), 'WAIT_INST/WAVE', mean(k, 'SQ_WAIT_INST_ANY') / max(wave, 1))
