cd "$(dirname "$0")/.."
OUT=gpurun_out/stale; mkdir -p $OUT
run() { name=$1; shift; env DBG_AMP=bf16 DBG_CR=2.0 DBG_POINTS=300000 DBG_SWEEPS=9 "$@" timeout -k 10 600 python tools/dbg_teacher_repro.py 120 360 640 > $OUT/$name.log 2>&1; echo "$name ($*): rc $? $(grep SUMMARY $OUT/$name.log)"; }
run bf16_full_base A=1
run bf16_full_cam0 U2MKD_CAMERA_STREAM=0
run bf16_full_tea0 U2MKD_TEACHER_STREAM=0
run bf16_full_both0 U2MKD_CAMERA_STREAM=0 U2MKD_TEACHER_STREAM=0
run bf16_full_both0_nodefer U2MKD_CAMERA_STREAM=0 U2MKD_TEACHER_STREAM=0 U2MKD_OVERLAP_WGRAD=0
