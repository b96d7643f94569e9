#!/bin/bash
cd /root/repo; R=$PWD
for m in 0 1 auto; do
  if [ $m = auto ]; then unset GDR_K7_PAIRS; else export GDR_K7_PAIRS=$m; fi
  rm -rf gpurun_out/tr
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr -o t -- python $R/bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-per-view-leg > /dev/null 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tr/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))))
k7=[(round((e-s)/1e3,1), "pairs" if "pairs" in n else "rows") for s,e,n in rows if "render_bwd" in n]
print("$m", k7)
PY
done
rm -rf gpurun_out/tr
