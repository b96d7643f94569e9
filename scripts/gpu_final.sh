#!/bin/bash
# End-of-round evidence run on the GPU box: tests, smoke, bench lines (with cpu_baseline), rocprofv3 kernel stats,
# PMC traffic, 2-rank smoke.  Everything lands in gpurun_out/final/ (copied into profiles/ by hand afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
TAG=${1:-r02}
bash scripts/gpu_check.sh > $O/check.log 2>&1; tail -3 $O/check.log
b() { name=$1; shift; timeout 1200 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_bench_$name.json")); r = d["roofline"] or {}
    print("$name", d["value"], "views/s", d["ms_per_step"], "ms/step D", d["config"]["num_rendered_per_view"], "dom", r.get("kernel"), r.get("frac"), "path", r.get("path_frac"), "cpu", (d["cpu_baseline"] or {}).get("value"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b c2 --workload c2
b c3 --workload c3
b c5 --workload c5
b c4_perview --per-view --unfused --no-cpu-baseline
b c4_torchloss --torch-loss --no-cpu-baseline
b c2_perview --workload c2 --per-view --unfused --no-cpu-baseline
b c5_perview --workload c5 --per-view --unfused --no-cpu-baseline
b c4_shell --layout shell --no-cpu-baseline
b c2_shell --workload c2 --layout shell --no-cpu-baseline
b c3_shell --workload c3 --layout shell --no-cpu-baseline
b c5_shell --workload c5 --layout shell --no-cpu-baseline
for wl in c4 c2 c5; do
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${wl}_kernel_stats.csv && python scripts/stats_print.py $f 3 8
  rm -rf $O/prof_$wl
done
for wl in c4 c2 c5; do
  BENCH_ARGS="--workload $wl" bash scripts/gpu_pmc.sh pmc_$wl > $O/pmc_$wl.log 2>&1
  cp gpurun_out/pmc_${wl}_summary.json $O/${TAG}_${wl}_pmc_summary.json 2>/dev/null; rm -rf gpurun_out/pmc_${wl}_[0-9]*
  tail -4 $O/pmc_$wl.log | cut -c1-300
done
echo "--- idle gaps (kernel timeline of bench steps)"
for wl in c4 c2 c3; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$wl -o t -- python $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace_$wl.log 2>&1)
  f=$(find $O/trace_$wl -name "*kernel_trace.csv" | head -1); python scripts/trace_gaps.py $f > $O/${TAG}_trace_gaps_$wl.txt; grep "^step" $O/${TAG}_trace_gaps_$wl.txt | head -3; rm -rf $O/trace_$wl
done
echo "--- abs-grad entry + device top-k"
python scripts/absgrad_bench.py 2>/dev/null | tee $O/${TAG}_absgrad.txt
echo "--- issue-rate microbenchmarks"
[ -x build/valu_rate ] && build/valu_rate > $O/${TAG}_valu_rate.txt && tail -9 $O/${TAG}_valu_rate.txt
[ -x build/valu_rate2 ] && build/valu_rate2 > $O/${TAG}_valu_rate2.txt
echo "--- 2 ranks on one GPU"
for wl in c2 c5; do for be in gloo; do  # (RCCL refuses two ranks on one device)
  timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl --single-device --no-roofline 2>/dev/null | tail -1 | tee $O/${TAG}_bench_2ranks_$wl.json | cut -c1-260
done; done
