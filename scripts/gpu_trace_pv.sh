#!/bin/bash
# per-kernel totals of the reference call pattern (--per-view --unfused): where does a drop-in user's time go
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
WL=${1:-c4}
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_pv -o t -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --per-view --unfused > $R/gpurun_out/trace_pv.log 2>&1)
f=$(find gpurun_out/trace_pv -name "*kernel_stats.csv" | head -1); python scripts/stats_print.py $f 3 30
f=$(find gpurun_out/trace_pv -name "*kernel_trace.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows)
busy=0; cs,ce=ev[0]
for s,e in ev[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("wall", (ev[-1][1]-ev[0][0])/1e3, "us busy", busy/1e3, "us", 100*busy/(ev[-1][1]-ev[0][0]), "%")
PY
rm -rf gpurun_out/trace_pv
