cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_x -o t -- python $R/bench.py --workload ${1:-c4} --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-per-view-leg > $R/gpurun_out/trace_x.log 2>&1)
f=$(find gpurun_out/trace_x -name "*kernel_trace.csv" | head -1); python scripts/chain_starts.py $f; rm -rf gpurun_out/trace_x
