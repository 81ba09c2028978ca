#!/bin/bash
# Round-5 discriminators for the stale-read item (NOTES N9): tools/dbg_teacher_repro.py (KD step from one state on one batch,
# consecutive teacher outputs compared) under one changed condition per run, plus the in-kernel probe.  Library BatchNorm2d and
# the dense pixel head put the most library work on the camera stream (the configuration with the highest deviation rate).
#   bash tools/stale_discriminators.sh [steps=150] [which...]      (on the GPU box; logs under gpurun_out/stale/)
cd "$(dirname "$0")/.."
STEPS=${1:-150}
shift
WHICH=${*:-probe base nocache serialize q4 q2 hashfill}
OUT=gpurun_out/stale
mkdir -p $OUT
export U2MKD_BN2D=0 U2MKD_SAMPLED_PIXEL_HEAD=0
run() {            # name, steps, env...
    local name=$1 steps=$2; shift 2
    echo "== $name ($*)"
    env "$@" timeout -k 10 420 python tools/dbg_teacher_repro.py $steps 360 640 > $OUT/$name.log 2>&1
    echo "   rc $? deviating comparisons: $(grep -c 'teacher rows differ' $OUT/$name.log | tr -d '\n') with rows > 0: $(grep 'teacher rows differ' $OUT/$name.log | grep -vc ': 0 of')  of $(grep -c '^step' $OUT/$name.log)"
}
for w in $WHICH; do
    case $w in
    probe) echo "== probe"; timeout -k 10 600 python tools/dbg_stale_probe.py $((STEPS + 50)) 360 640 > $OUT/probe.log 2>&1; echo "   rc $?"; tail -n 14 $OUT/probe.log;;
    base) run base $STEPS A=1;;
    nocache) run nocache 40 PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1;;
    serialize) run serialize $STEPS AMD_SERIALIZE_KERNEL=3;;
    blocking) run blocking 60 HIP_LAUNCH_BLOCKING=1;;
    q4) run q4 $STEPS GPU_MAX_HW_QUEUES=4;;
    q2) run q2 $STEPS GPU_MAX_HW_QUEUES=2;;
    q1) run q1 $STEPS GPU_MAX_HW_QUEUES=1;;
    hashfill) run hashfill $STEPS U2MKD_DEBUG_HASH_FILL=1;;
    esac
done
