"""Kernel families of the TIMED steps only, from a rocprofv3 kernel trace (csv or csv.gz): the last `steps * ms_per_step`
of the trace (MIOpen's first-call search kernels and graph captures of the warm-up stay out).
usage: python tools/trace_steps.py <kernel_trace.csv[.gz]> <steps> <ms_per_step> [top]"""
import collections
import csv
import gzip
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from kernel_groups import grp  # noqa: E402

path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
op = gzip.open if path.endswith('.gz') else open
rows = list(csv.DictReader(op(path, 'rt')))
end = max(int(r['End_Timestamp']) for r in rows)
t0 = end - steps * ms * 1e6
sel = [r for r in rows if int(r['Start_Timestamp']) >= t0]
acc = collections.defaultdict(lambda: [0.0, 0])
byname = collections.defaultdict(lambda: [0.0, 0])
for r in sel:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    g = grp(r['Kernel_Name'])
    acc[g][0] += d; acc[g][1] += 1
    byname[r['Kernel_Name']][0] += d; byname[r['Kernel_Name']][1] += 1
tot = sum(v[0] for v in acc.values())
print('timed window: %d launches = %.0f per step, kernel time %.1f ms per step' % (len(sel), len(sel) / steps, tot / 1e6 / steps))
for g, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
    print('%-20s %7.2f ms/step %6d launches/step' % (g, t / 1e6 / steps, c / steps))
print()
for n, (t, c) in sorted(byname.items(), key=lambda x: -x[1][0])[:top]:
    print('%8.2f ms/step %6.1f  %s' % (t / 1e6 / steps, c / steps, n[:110]))
