#!/bin/bash
# per-dispatch durations of the roofline leg's kernels (rocprofv3 --kernel-trace of `bench.py --kernel-only`), in launch order
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/trace_ko
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o ko -- python3 "$REPO/bench.py" --kernel-only > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" || echo "rc=$?"
F=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'conv_tp_kernel<4, 1, 64, false, 2' in r['Kernel_Name'] or 'conv_wgrad_x3_kernel<false>' in r['Kernel_Name']]
# runs of the same kernel
runs, cur = [], None
for n, d in seq:
    tag = 'tp' if 'conv_tp' in n else 'wg'
    if cur is None or cur[0] != tag:
        cur = [tag, []]; runs.append(cur)
    cur[1].append(d)
for tag, ds in runs:
    ds2 = sorted(ds)
    print('%s x%d  mean %.1f us  median %.1f  min %.1f  max %.1f' % (tag, len(ds), sum(ds) / len(ds) / 1e3, ds2[len(ds) // 2] / 1e3, ds2[0] / 1e3, ds2[-1] / 1e3))
P
