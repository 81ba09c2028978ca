U2MKD_WGRAD_X3=3 python -m pytest tests/test_gpu_torchsparse_ops.py -x -q -k "subm_conv_fwd_bwd or strided_and or both_conv" > gpurun_out/e35_test.log 2>&1; echo "test x3=3 rc=$?"; tail -2 gpurun_out/e35_test.log | cut -c1-200
for m in 2 3 2 3; do
U2MKD_WGRAD_X3=$m timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/e35_kd$m.log 2> gpurun_out/e35_kd$m.err; tail -1 gpurun_out/e35_kd$m.log | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('WGRAD_X3=$m KD', r['ms_per_step'], {k:v.get('ms_per_step') for k,v in r['secondary'].items()})"
done
