#!/bin/bash
# KD step by (hardware queues, geometry mode, N>1 path): one line each, written as they finish.
#   tools/ab_modes.sh "Q STAGED DDP" ...      e.g.  tools/ab_modes.sh "4 1 0" "8 0 1"   (STAGED = U2MKD_STAGED_GEOMETRY)
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  extra=""; [ "$3" = "1" ] && extra="U2MKD_FORCE_DDP=1 U2MKD_FORCE_SYNC_BN=1"
  env $extra GPU_MAX_HW_QUEUES=$1 U2MKD_STAGED_GEOMETRY=$2 python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues %s geometry %-5s ddp-path %s: mean %.2f median %.2f host %.1f deviating %s' % (sys.argv[1], sys.argv[2], sys.argv[3], d['ms_per_step'], d['ms_per_step_median'], d['config']['host_issue_ms_per_step'], d['config']['teacher_deviating_steps']))" $1 $2 $3 | tee -a gpurun_out/ab_modes.txt
done
