"""Per-stream view of a rocprofv3 kernel trace (csv or csv.gz) of the KD step: busy time and launches per stream over
a window at the end of the run, the top kernels of each stream, and the largest idle gaps of the busiest (main)
stream with the kernels either side and what the other streams ran meanwhile.
usage: stream_timeline.py <kernel_trace.csv[.gz]> [window_ms=800] [tail_skip_ms=100] [steps_in_window]"""
import collections, csv, gzip, re, sys

path = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
opener = gzip.open if path.endswith('.gz') else open
rows = []
with opener(path, 'rt') as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id') or r.get('Queue_Id')))
rows.sort()
end = rows[-1][1]
lo, hi = end - int((win + skip) * 1e6), end - int(skip * 1e6)
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = hi - lo
steps = float(sys.argv[4]) if len(sys.argv) > 4 else None


def union(rs):
    b, cs, ce = 0, rs[0][0], rs[0][1]
    for s, e, *_ in rs[1:]:
        if s > ce:
            b += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return b + ce - cs


def short(n):
    n = re.sub(r'ROCPRIM_\d+_NS::', '', n).replace('void ', '').replace('at::native::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:64]


print('window %.0f ms, %d launches, GPU busy (union of all streams) %.1f %%, sum of kernel time %.1f %% of the window'
      % (span / 1e6, len(rows), 100 * union(rows) / span, 100 * sum(e - s for s, e, *_ in rows) / span))
per = collections.defaultdict(list)
for r in rows:
    per[r[3]].append(r)
order = sorted(per, key=lambda q: -sum(e - s for s, e, *_ in per[q]))
for q in order[:4]:
    rs = per[q]
    print('stream %s: %d launches, busy %.1f %% of the window' % (q, len(rs), 100 * union(rs) / span))
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _ in rs:
        agg[short(n)][0] += e - s
        agg[short(n)][1] += 1
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
        per_step = (' %6.2f ms/step' % (t / 1e6 / steps)) if steps else ''
        print('     %-64s %7.2f ms %6d calls %7.1f us%s' % (k, t / 1e6, c, t / c / 1e3, per_step))
main = per[order[0]]
gaps = []
for (s0, e0, n0, _), (s1, e1, n1, _) in zip(main, main[1:]):
    if s1 - e0 > 150_000:
        gaps.append((s1 - e0, e0, s1, n0, n1))
tot = sum(g[0] for g in gaps)
print('main stream %s: %d gaps > 150 us totalling %.1f ms (%.1f %% of the window); the 25 largest:' % (order[0], len(gaps), tot / 1e6, 100 * tot / span))
others = [r for r in rows if r[3] != order[0]]
for g in sorted(gaps, reverse=True)[:25]:
    mid = [r for r in others if r[1] > g[1] and r[0] < g[2]]
    busy = union(sorted(mid)) if mid else 0
    names = collections.Counter(short(r[2])[:28] for r in mid).most_common(2)
    print('  %7.1f us  after %-34s before %-34s | other streams busy %3.0f %%: %s'
          % (g[0] / 1e3, short(g[3])[:34], short(g[4])[:34], 100 * min(busy, g[0]) / g[0], ', '.join('%s x%d' % kv for kv in names)))
