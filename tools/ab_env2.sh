#!/bin/bash
# same-box A/B of environment settings on the KD step with the per-run median and extremes (steps of 20):
#   tools/ab_env2.sh "VAR=a [VAR2=..]" "VAR=b [..]" [rounds=4]
A=$1; B=$2; N=${3:-4}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('AB %-60s mean %.2f median %.2f min/max %s host %.1f' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], d['config']['host_issue_ms_per_step']))" "$v"
  done
done
