#!/bin/bash
# PMC counters of the roofline leg (bench.py --kernel-only); separate passes per counter group.
# usage: tools/pmc_kernel_only.sh <tag>
TAG=${1:-pmc}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  local out=$REPO/gpurun_out/pmc_${TAG}_$name
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -o pmc -- python3 "$REPO/bench.py" --kernel-only > "$out/stdout.txt" 2> "$out/stderr.txt" || echo "rc=$? for $name"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $REPO/gpurun_out/pmc_${TAG}_*
