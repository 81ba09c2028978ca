#!/bin/bash
# PMC counters of the sptr attention kernels inside one KD step (tools/sptr_step_times.py); two passes.
TAG=${1:-sptr}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  local out=$REPO/gpurun_out/pmc_${TAG}_$name
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -o pmc -- python3 "$REPO/tools/sptr_step_times.py" > "$out/stdout.txt" 2> "$out/stderr.txt" || echo "rc=$? for $name"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'sptr_' in r['Kernel_Name'] and 'delta' not in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
# the LAST dispatches = the measured step; group by (kernel, grid)
acc = collections.defaultdict(dict)
for r in rows:
    k = (r['Kernel_Name'].split('(')[0][-40:], r.get('Grid_Size', r.get('Grid_Size_X', '')))
    acc[k][r['Counter_Name']] = float(r['Counter_Value'])
for k, d in acc.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
  find "$out" -name '*.csv' -size +1M -delete      # (the raw counter tables are tens of MB: keep the summary only)
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
