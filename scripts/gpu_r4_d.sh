#!/bin/bash
# round 4, run D: whole GPU suite after the surfel-group fixes; surfel parity statistics at tighter floors; new bench modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
GDR_TEST_STATS=1 timeout 1200 python -m pytest tests/test_gpu_oracle_fullsize.py tests/test_gpu_surfel.py -q -rP -k "surfel" 2>&1 | grep -E "^\[|passed|failed" | cut -c1-200 > $O/surfel_stats.txt; tail -1 $O/surfel_stats.txt
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || tail -5 $O/bench_$name.err; python - <<PY
import json
try:
    d = json.load(open("$O/bench_$name.json")); r = d.get("roofline") or {}; pv = d.get("per_view") or {}
    print("$name", d["value"], d["unit"], d["ms_per_step"], "| per_view", pv.get("value"), "| dom", r.get("kernel"), r.get("frac"), "path", r.get("path_frac"), r.get("path_frac_built"), r.get("path_frac_measured"), "| comm", d.get("comm_ms"), "| flips", (d.get("psnr_vs_oracle") or {}).get("threshold_flips"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b c2_fwd --workload c2 --forward-only
b c3_fwd --workload c3 --forward-only
b c4_fwd --forward-only
b c5_fwd --workload c5 --forward-only
b c3step --workload c3step --steps 6 --warmup 2
b rccl1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce
b rccl1_keep --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce --keep-grads
python scripts/host_split.py 2>/dev/null | tail -1 | tee $O/host_split.txt
