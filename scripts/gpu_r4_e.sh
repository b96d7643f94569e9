#!/bin/bash
# round 4, run E: K6 of all views in one launch (K.K6_VIEWS) A/B, no_grad fast path, launch-mode parity test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_surfel.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"; }
for wl in c4 c3 c2; do for rep in 1 2; do for m in 0 1 2; do echo -n "$wl K6_VIEWS=$m rep$rep: "; GDR_K6_VIEWS=$m b --workload $wl; done; done; done 2>&1 | tee $O/ab.txt
for wl in c2 c3 c4 c5; do echo -n "$wl fwd-only: "; python bench.py --workload $wl --forward-only --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], 'per_view', d['per_view']['value'])"; done | tee -a $O/ab.txt
bash scripts/gpu_timeline.sh c2 --no-per-view-leg > /dev/null 2>&1; cp gpurun_out/timeline_c2.txt $O/; head -60 $O/timeline_c2.txt
