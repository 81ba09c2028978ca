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
