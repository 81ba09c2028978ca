#!/bin/bash
# What the N>1 code path costs at ONE rank, piece by piece (no communication partner: launches and host work only):
#   plain | gradient reducer only | reducer + synchronising BatchNorm kernels; and the kernel count of the last one.
set -u
out=gpurun_out/ab_syncbn; mkdir -p $out
B="python bench.py --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline"
$B > $out/plain.json 2> $out/plain.err
U2MKD_FORCE_DDP=1 $B > $out/ddp.json 2> $out/ddp.err
U2MKD_FORCE_DDP=1 U2MKD_FORCE_SYNC_BN=1 $B > $out/ddp_sync.json 2> $out/ddp_sync.err
for f in plain ddp ddp_sync; do
  python - $out/$f.json $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:9s} {d['ms_per_step']:.2f} ms/step (median {d.get('ms_per_step_median')}), host issue {d.get('config', {}).get('host_issue_ms_per_step')}")
PY
done
# kernel count and busy time of the last variant (2 warm-up + 6 timed steps under the tracer)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
for v in ddp ddp_sync; do
  e=""; [ $v = ddp_sync ] && e="1"
  U2MKD_FORCE_DDP=1 U2MKD_FORCE_SYNC_BN=${e:-0} rocprofv3 --kernel-trace --stats -d $out/prof_$v -o t -- python bench.py --steps 6 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline > $out/prof_$v.json 2> $out/prof_$v.err
  python - $out/prof_$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if not f: print('no stats'); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
calls = sum(int(r['Calls']) for r in rows); ns = sum(float(r['TotalDurationNs']) for r in rows)
print(f"{sys.argv[2]}: {calls} kernel launches / 9 steps = {calls/9:.0f} per step, busy {ns/9e6:.1f} ms per step")
for r in sorted(rows, key=lambda r: -int(r['Calls']))[:14]:
    print(f"   {int(r['Calls'])/9:7.1f}/step  {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:90]}")
PY
done
