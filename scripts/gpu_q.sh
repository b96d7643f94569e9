#!/bin/bash
O=gpurun_out/q; mkdir -p $O; R=$(pwd)
wl=c4
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_pv -o t -- python $R/bench.py --workload $wl --per-view --unfused --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/trace_pv.log 2>&1)
f=$(find $O/trace_pv -name "*kernel_trace.csv" | head -1); python scripts/trace_gaps.py $f --every 4 --timeline > $O/timeline_pv.txt; rm -rf $O/trace_pv
timeout 300 python bench.py --workload c4 --per-view --unfused --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --host-profile 2> $O/host_pv.txt >/dev/null
