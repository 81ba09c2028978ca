#!/bin/bash
# Same-box A/B of the two bf16 matrix-instruction forms (NOTES N9): the shipped library (two v_mfma_f32_16x16x16_bf16 per K = 32
# product) against a variant built with -DU2MKD_MFMA_GFX950_K32=1 (gfx950's v_mfma_f32_16x16x32_bf16): roofline leg, KD step time,
# and the teacher's deviating steps in both.   bash tools/ab_mfma_form.sh   (GPU box; results: gpurun_out/ab_mfma/)
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab_mfma; mkdir -p $OUT
bash tools/build_variant.sh _k32 "-DU2MKD_MFMA_GFX950_K32=1" conv_tp.hip conv_px3.hip conv_wgrad_x3.hip > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
for rep in 1 2; do
  for suf in "" _k32; do
    U2MKD_LIB_SUFFIX=$suf U2MKD_BENCH_WIDE=1 python bench.py --kernel-only > $OUT/ko${suf}_$rep.json 2> /dev/null
    U2MKD_LIB_SUFFIX=$suf python bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 30 --warmup 6 > $OUT/kd${suf}_$rep.json 2> /dev/null
  done
done
for suf in "" _k32; do
  echo "== tests/test_gpu_concurrent_streams.py with libu2mkd_hip$suf.so"
  U2MKD_LIB_SUFFIX=$suf python -m pytest tests/test_gpu_concurrent_streams.py -x -q 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tail -4
done
python - <<'PY'
import json, glob
for suf, name in (('', 'shipped: 2 x v_mfma_f32_16x16x16_bf16'), ('_k32', 'variant: v_mfma_f32_16x16x32_bf16')):
    for rep in (1, 2):
        ko = json.load(open('gpurun_out/ab_mfma/ko%s_%d.json' % (suf, rep)))
        kd = json.load(open('gpurun_out/ab_mfma/kd%s_%d.json' % (suf, rep)))
        r, w = ko['roofline'], ko.get('roofline_wide', {})
        print('%-42s run %d: SubMConv3d 64->64 group %.1f us warm (frac %.3f) / %.1f us cold (frac %.3f); 512->512 group %s ms (%s TF fp32-eq); '
              'KD step %.2f ms (median %.2f), teacher deviating steps %s of %s'
              % (name, rep, r['warm']['ms']['total'] * 1e3, r['warm']['frac'], r['ms']['total'] * 1e3, r['frac'], w.get('ms', {}).get('total'), w.get('achieved'),
                 kd['ms_per_step'], kd['ms_per_step_median'], kd['config']['teacher_deviating_steps'], kd['config']['teacher_steps_compared']))
PY
