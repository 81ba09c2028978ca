#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summary -> gpurun_out/prof_<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-run}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 "$REPO/bench.py" "$@" > "$OUT/bench_stdout.txt" 2> "$OUT/bench_stderr.txt" || echo "rocprofv3 rc=$?"
ls -R "$OUT" | head -30
