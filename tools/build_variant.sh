#!/bin/bash
# Build an alternative libu2mkd_hip<suffix>.so with extra -D flags on selected translation units (same-box A/B, tools/ab_variants.py).
# usage: tools/build_variant.sh <suffix> "<flags>" file1.hip [file2.hip ...]
set -e
cd "$(dirname "$0")/.."
SUF=$1; FLAGS=$2; shift 2
python -m u2mkd_amd.build > /dev/null
OBJS=""
for o in build/obj/*.o; do
  b=$(basename $o .o)
  case "$b" in *__v_*) continue;; esac
  skip=0
  for f in "$@"; do [ "$b" = "$(basename $f .hip)" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $FLAGS -c u2mkd_amd/csrc/$b.hip -o build/obj/${b}__v_${SUF}.o &
done
wait
for f in "$@"; do b=$(basename $f .hip); OBJS="$OBJS build/obj/${b}__v_${SUF}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o u2mkd_amd/lib/libu2mkd_hip${SUF}.so $OBJS
echo "built u2mkd_amd/lib/libu2mkd_hip${SUF}.so ($FLAGS)"
