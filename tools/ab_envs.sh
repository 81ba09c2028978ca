#!/bin/bash
# same-box comparison of several environment settings of the KD step: tools/ab_envs.sh ROUNDS "VAR=a" "VAR=b VAR2=c" ...
# (one bare KD leg per setting and round, K = 20 / W = 6; prints mean, median and the teacher's deviating steps)
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    r=$(env $v python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"ms_per_step_median": [0-9.]*\|"teacher_deviating_steps": [0-9]*\|"host_issue_ms_per_step": [0-9.]*' | tr '\n' ' ')
    echo "AB [$v] $r"
  done
done
