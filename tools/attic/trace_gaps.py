"""GPU busy / idle time from a rocprofv3 kernel trace (csv): union of the kernel intervals over all streams vs the
wall span, per-step figures, and the largest idle gaps with the kernels either side (host-bound sections).
usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id') or r.get('Queue_Id')))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort()
# keep the last 60 % of the trace (timed region, after warm-up)
t_lo = rows[0][0] + int(0.4 * (rows[-1][1] - rows[0][0]))
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
prev_name = rows[0][2]
for s, e, nme, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev_name, nme))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = nme
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _, _ in rows)
per = {}
for s, e, _, q in rows:
    per[q] = per.get(q, 0) + e - s
print('kernel time per stream/queue (ms): ' + ', '.join(f'{q}: {v/1e6:.1f}' for q, v in sorted(per.items(), key=lambda kv: -kv[1])))
print(f'span {span/1e6:.1f} ms, GPU busy (union) {busy/1e6:.1f} ms = {100*busy/span:.1f} %, sum of kernel durations {ksum/1e6:.1f} ms, launches {len(rows)}')
print(f'idle {100*(span-busy)/span:.1f} % in {len(gaps)} gaps; gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.1f} ms; '
      f'gaps <= 20 us: {sum(g[0] for g in gaps if g[0] <= 20000)/1e6:.1f} ms')
for g in sorted(gaps, reverse=True)[:25]:
    print(f'{g[0]/1e3:8.1f} us  after {g[1][:60]:60s} before {g[2][:60]}')
