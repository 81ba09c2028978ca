#!/bin/bash
# The default configuration's acceptance run for the frozen teacher: consecutive KD steps from one state on
# one batch, teacher logits compared bit for bit -- 520 steps at 6 x 360x640, 520 at 6 x 900x1600, 200 under bf16 autocast, and the
# library-BatchNorm / dense-head configuration that showed the highest rate before the fix.   (GPU box; logs: gpurun_out/stale/)
cd "$(dirname "$0")/.."
OUT=gpurun_out/stale
mkdir -p $OUT
N=${1:-520}
run() { name=$1; shift; env "$@" timeout -k 10 900 python tools/dbg_teacher_repro.py $STEPS $SIZE > $OUT/$name.log 2>&1; echo "$name ($*): rc $? $(grep SUMMARY $OUT/$name.log)"; }
STEPS=$N SIZE="360 640" run final_360 A=1
STEPS=$N SIZE="900 1600" run final_900 A=1
STEPS=200 SIZE="360 640" run final_bf16 DBG_AMP=bf16 DBG_CR=2.0
STEPS=300 SIZE="360 640" run final_libbn U2MKD_BN2D=0 U2MKD_SAMPLED_PIXEL_HEAD=0
