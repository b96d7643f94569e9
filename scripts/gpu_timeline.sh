#!/bin/bash
# kernel-by-kernel timeline of one bench step: bash scripts/gpu_timeline.sh c4 [extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
WL=${1:-c4}; shift
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$WL -o t -- python $R/bench.py --workload $WL --steps 4 --warmup 13 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/trace_$WL.log 2>&1)
f=$(find gpurun_out/trace_$WL -name "*kernel_trace.csv" | head -1); python scripts/trace_gaps.py $f --timeline > gpurun_out/timeline_$WL.txt; head -4 gpurun_out/timeline_$WL.txt; rm -rf gpurun_out/trace_$WL
