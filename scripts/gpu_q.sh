#!/bin/bash
O=gpurun_out/q; mkdir -p $O; R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q 2>&1 | tail -8
for wl in c2 c3 c4; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$wl -o t -- python $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/$O/trace_$wl.log 2>&1)
  f=$(find $O/trace_$wl -name "*kernel_trace.csv" | head -1); python scripts/trace_gaps.py $f > $O/trace_gaps_$wl.txt; grep "^step" $O/trace_gaps_$wl.txt | head -6; rm -rf $O/trace_$wl
done
