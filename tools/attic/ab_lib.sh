#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <suffix of the alternative .so> -- runs tools/exp_step_bounds.py on each
cd "$(dirname "$0")/.."
for v in "" "$1"; do
  python - <<PY 2>&1 | grep VARIANT | sed "s/baseline/lib${v:-(current)}/"
import sys; sys.path.insert(0, '.')
import u2mkd_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace('libu2mkd_hip.so', 'libu2mkd_hip${v}.so')
sys.argv = ['exp_step_bounds']
import runpy; runpy.run_path('tools/exp_step_bounds.py', run_name='__main__')
PY
done
