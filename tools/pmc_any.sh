#!/bin/bash
# One rocprofv3 --pmc pass over any tool script; per-kernel means of the counters for kernels matching a pattern.
# usage: tools/pmc_any.sh <tag> <kernel-substring> "<counters>" <script> [args...]      (the program after -- is python3 itself)
TAG=$1; PAT=$2; CNT=$3; shift 3
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT" -o pmc -- python3 "$REPO/$1" "${@:2}" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" || echo "rc=$?"
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][:70]
    if sys.argv[2] not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}, 'dispatches', max(cnt[(k, c)] for c in d))
PY
find "$OUT" -name '*kernel_trace.csv' -delete
