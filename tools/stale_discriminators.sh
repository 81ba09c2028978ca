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
    env "$@" timeout -k 10 600 python tools/dbg_teacher_repro.py $steps ${SIZE:-360 640} > $OUT/$name.log 2>&1
    echo "   rc $? deviating comparisons: $(grep -c 'teacher rows differ' $OUT/$name.log | tr -d '\n') with rows > 0: $(grep 'teacher rows differ' $OUT/$name.log | grep -vc ': 0 of')  of $(grep -c '^step' $OUT/$name.log)"
}
for w in $WHICH; do
    case $w in
    probeA) echo "== probe, student after teacher"; U2MKD_DEBUG_ORDER=student_after_teacher timeout -k 10 600 python tools/dbg_stale_probe.py $STEPS 360 640 > $OUT/probeA.log 2>&1; echo "   rc $?"; grep -v "^step\|^ \|MIOpen" $OUT/probeA.log | tail -n 6;;
    probeB) echo "== probe, teacher after camera head"; U2MKD_DEBUG_ORDER=teacher_after_camera timeout -k 10 600 python tools/dbg_stale_probe.py $STEPS 360 640 > $OUT/probeB.log 2>&1; echo "   rc $?"; grep -v "^step\|^ \|MIOpen" $OUT/probeB.log | tail -n 6;;
    probeC) echo "== probe, teacher alone on the GPU"; U2MKD_DEBUG_ORDER=teacher_after_camera,student_after_teacher timeout -k 10 600 python tools/dbg_stale_probe.py $STEPS 360 640 > $OUT/probeC.log 2>&1; echo "   rc $?"; grep -v "^step\|^ \|MIOpen" $OUT/probeC.log | tail -n 6;;
    probeD) echo "== probe, no deferred weight gradients"; U2MKD_OVERLAP_WGRAD=0 timeout -k 10 600 python tools/dbg_stale_probe.py $STEPS 360 640 > $OUT/probeD.log 2>&1; echo "   rc $?"; grep -v "^step\|^ \|MIOpen" $OUT/probeD.log | tail -n 6;;
    probe) echo "== probe"; timeout -k 10 600 python tools/dbg_stale_probe.py $((STEPS + 50)) 360 640 > $OUT/probe.log 2>&1; echo "   rc $?"; grep -v "MIOpen" $OUT/probe.log | tail -n 40;;
    base) run base $STEPS A=1;;
    nocache) run nocache 40 PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1;;
    serialize) run serialize $STEPS AMD_SERIALIZE_KERNEL=3;;
    blocking) run blocking 60 HIP_LAUNCH_BLOCKING=1;;
    q4) run q4 $STEPS GPU_MAX_HW_QUEUES=4;;
    q2) run q2 $STEPS GPU_MAX_HW_QUEUES=2;;
    q1) run q1 $STEPS GPU_MAX_HW_QUEUES=1;;
    q5) run q5 $STEPS GPU_MAX_HW_QUEUES=5;;
    q6) run q6 $STEPS GPU_MAX_HW_QUEUES=6;;
    q4long) run q4long 500 GPU_MAX_HW_QUEUES=4;;
    q4big) SIZE="900 1600" run q4big 150 GPU_MAX_HW_QUEUES=4 U2MKD_SAMPLED_PIXEL_HEAD=1;;
    q8big) SIZE="900 1600" run q8big 100 GPU_MAX_HW_QUEUES=8 U2MKD_SAMPLED_PIXEL_HEAD=1;;
    kahdp) run kahdp $STEPS DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1;;
    devkarg0) run devkarg0 $STEPS HIP_FORCE_DEV_KERNARG=0;;
    devkarg1) run devkarg1 $STEPS HIP_FORCE_DEV_KERNARG=1;;
    kpool) run kpool $STEPS HSA_KERNARG_POOL_SIZE=67108864;;
    fgs0) run fgs0 $STEPS ROC_USE_FGS_KERNARG=0;;
    argcopy) run argcopy $STEPS DEBUG_HIP_KERNARG_COPY_OPT=0;;
    map) for q in 2 4 5 6 8; do GPU_MAX_HW_QUEUES=$q python tools/queue_map.py > $OUT/map_q$q.log 2>&1; cat $OUT/map_q$q.log; done;;
    hashfill) run hashfill $STEPS U2MKD_DEBUG_HASH_FILL=1;;
    esac
done
