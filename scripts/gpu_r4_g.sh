#!/bin/bash
# round 4, run G: native multi-view forward (gdr_forward_views) — whole suite + benches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'per_view', (d.get('per_view') or {}).get('value'))"; }
for wl in c4 c3 c2 c5; do for rep in 1 2; do echo -n "$wl rep$rep: "; b --workload $wl; done; done 2>&1 | tee $O/bench.txt
echo -n "c3step: "; python bench.py --workload c3step --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'per_view', (d.get('per_view') or {}).get('value'))" | tee -a $O/bench.txt
for wl in c4 c2 ; do echo -n "$wl shell: "; b --workload $wl --layout shell --no-per-view-leg; done | tee -a $O/bench.txt
bash scripts/gpu_timeline.sh c2 --no-per-view-leg > /dev/null 2>&1; cp gpurun_out/timeline_c2.txt $O/; grep "^step" $O/timeline_c2.txt
