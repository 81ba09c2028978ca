for g in 1024 997 960 900 741 1480; do
U2MKD_WGRAD_GMAX=$g timeout -k 10 200 python bench.py --kernel-only > gpurun_out/e2_g$g.log 2>&1; tail -1 gpurun_out/e2_g$g.log | python -c "import json,sys; r=json.loads(sys.stdin.read())['roofline']; print('GMAX $g', r['ms'], r['frac'], r['cold']['ms'])"
done
