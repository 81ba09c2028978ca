"""The configs[4] leg of bench.py alone (multi-sweep teacher-only step, bf16 storage, 300 000 points, cr 2.0) -- same-box A/B of
library variants (U2MKD_LIB_SUFFIX) and switches.   python tools/time_configs4.py [steps=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sys.argv = sys.argv[:1]
import bench
args = bench.parse()
s, n, d = bench.build_step(args, 0, 'kd', args.image_hw, sweeps=9, dtype='bf16', voxels=300000, cr=2.0, cr_t=2.0)
dt, _ = bench.timed_run(s, 5, steps, 1)
print('configs4 leg: %.2f ms per step (median %.2f); %s' % (dt / steps * 1e3, bench.timed_run.median_ms, d[:90]))
