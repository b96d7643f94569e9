#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats (CSV) of a short default-workload bench run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r01}
ARGS=${2:---steps 3 --warmup 1 --no-cpu-baseline --no-roofline}
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*stats*" | head; 
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
