#!/bin/bash
# one GPU call: default bench line, rocprof stats of the KD step and of the roofline leg, PMC passes; -> gpurun_out/
# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>
TAG=${1:-r2b}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_$TAG.json
bash tools/profile_bench.sh $TAG --no-secondary --no-cpu-baseline --steps 10 --warmup 4 > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
bash tools/profile_bench.sh ${TAG}_ko --kernel-only > gpurun_out/prof_${TAG}_ko.log 2>&1; echo "prof ko rc=$?"
bash tools/pmc_kernel_only.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc=$?"
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv gpurun_out/prof_${TAG}_ko/*kernel_trace.csv gpurun_out/pmc_${TAG}_*/*kernel_trace.csv
du -sh gpurun_out | tail -1
