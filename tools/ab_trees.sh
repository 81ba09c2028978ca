#!/bin/bash
# same-box A/B of two source trees of the package (an older tree unpacked under .ab_old/ by `git archive <commit> u2mkd_amd bench.py`):
# the KD step and the same step on the N>1 code path, two rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/ab
for i in 1 2; do
for t in .ab_old .; do
  cd $R/$t
  n=$(echo $t | tr -d ./)_$i
  python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/ab/kd_$n.json 2> $R/gpurun_out/ab/kd_$n.err
  U2MKD_FORCE_DDP=1 python bench.py --no-secondary --no-roofline --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/ab/ddp_$n.json 2> $R/gpurun_out/ab/ddp_$n.err
  echo "TREE $t  kd $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/ab/kd_$n.json)  ddp-path $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/ab/ddp_$n.json)"
done
done
tail -3 $R/gpurun_out/ab/*.err | tail -30
