set -e
B="python bench.py --no-secondary --no-roofline --no-cpu-baseline"
for v in "0 3 10" "1 3 10" "0 6 20" "1 6 20" "1 3 10" "0 3 10"; do
  set -- $v
  if [ "$1" = 1 ]; then export U2MKD_FORCE_DDP=1; else unset U2MKD_FORCE_DDP; fi
  $B --warmup $2 --steps $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddp=$1 warm=$2 steps=$3', d['ms_per_step'])"
done
