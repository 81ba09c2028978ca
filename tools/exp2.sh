timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/e19_b.log 2> gpurun_out/e19_b.err; echo rc=$?; tail -1 gpurun_out/e19_b.log | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('KD', r['ms_per_step'], r['value'])
for k,v in r['secondary'].items(): print(k, v.get('ms_per_step'), v.get('value'), v.get('error'), v.get('workload','')[:160])"; grep "bench " gpurun_out/e19_b.err | tail -8
