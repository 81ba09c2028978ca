#!/bin/bash
# same-box A/B of one environment switch on the KD step: tools/ab_env.sh VAR=a VAR=b [rounds=3]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    ms=$(env $v python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "AB $v $ms"
  done
done
