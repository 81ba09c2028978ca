#!/bin/bash
# kernel trace of the bare KD step (no secondaries / roofline / CPU baseline) + per-stream timeline of its last steps
# usage: tools/step_trace.sh <tag> [steps=12]
set -e
TAG=${1:-run}; STEPS=${2:-12}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o step -- python3 "$REPO/bench.py" --no-secondary --no-roofline --no-cpu-baseline --steps "$STEPS" --warmup 4 > "$OUT/bench_stdout.txt" 2> "$OUT/bench_stderr.txt" || echo "rocprofv3 rc=$?"
F=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
python3 "$REPO/tools/stream_timeline.py" "$F" 600 50 > "$OUT/timeline.txt" 2>&1 || true
# the window = the last 8 steps AT THE STEP TIME THIS RUN MEASURED (the profiler slows the host: ~107 instead of ~70 ms per step)
MS=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench_stdout.txt').read().strip().splitlines()[-1])['ms_per_step'])")
{ echo "under rocprofv3 --kernel-trace: $MS ms per step (the profiler slows the host; kernel durations are the unprofiled ones, inflated by the concurrency of the streams)"; python3 "$REPO/tools/trace_steps.py" "$F" 8 "$MS" 60; } > "$OUT/steps.txt" 2>&1 || true
gzip -f "$F"
find "$OUT" -name '*.csv.gz' -size +40M -delete
ls -la "$OUT" | head
