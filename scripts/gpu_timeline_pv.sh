#!/bin/bash
# kernel-by-kernel timeline of one step of the UNCHANGED caller's loop (render_img per view, one backward):
#   bash scripts/gpu_timeline_pv.sh c4   -> gpurun_out/timeline_pv_c4.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
WL=${1:-c4}; shift
V=$(python -c "import bench; print(bench.WORKLOADS['$WL']['views_per_gpu'])")
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pv_$WL -o t -- python $R/bench.py --workload $WL --steps 4 --warmup 13 --no-cpu-baseline --no-roofline --per-view --unfused "$@" > $R/gpurun_out/trace_pv_$WL.log 2>&1)
f=$(find gpurun_out/trace_pv_$WL -name "*kernel_trace.csv" | head -1); python scripts/trace_gaps.py $f --every $V --timeline > gpurun_out/timeline_pv_$WL.txt; head -30 gpurun_out/timeline_pv_$WL.txt; rm -rf gpurun_out/trace_pv_$WL
