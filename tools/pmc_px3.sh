#!/bin/bash
# PMC counters of the wide-layer pair kernel (tools/ab_px3.py on two shapes); separate passes per counter group.
# usage: tools/pmc_px3.sh <tag>
TAG=${1:-px3}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  local out=$REPO/gpurun_out/pmc_${TAG}_$name
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -o pmc -- python3 "$REPO/tools/ab_px3.py" 4,256,256 1,96,96 > "$out/stdout.txt" 2> "$out/stderr.txt" || echo "rc=$? for $name"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'px3' not in k and 'gather_sum' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: round(v / cnt[(k, c)], 1) for c, v in d.items()})
PY
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD
