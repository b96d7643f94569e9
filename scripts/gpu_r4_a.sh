#!/bin/bash
# round 4, run A: the one-launch K7 (GDR_K7_VIEWS) — parity tests, then same-box A/B against per-view launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "k7_of_all_views or fused_multiview or loss_folded or screenspace_absgrad or cut_tile" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"; }
for wl in c4 c3 c2; do
  for rep in 1 2; do
    for mode in 0 1 2; do echo -n "$wl K7_VIEWS=$mode rep$rep: "; GDR_K7_VIEWS=$mode b --workload $wl; done
  done
done 2>&1 | tee $O/ab.txt
for wl in c4 c3 c2; do for mode in 0 1; do
  echo -n "$wl noslp K7_VIEWS=$mode: "; GDR_LIB_PATH=$PWD/generativedensification_amd/lib/variants/libgdr_noslp.so GDR_K7_VIEWS=$mode b --workload $wl; done; done 2>&1 | tee -a $O/ab.txt
for wl in c4 c2; do for mode in 0 1; do
  echo -n "$wl shell K7_VIEWS=$mode: "; GDR_K7_VIEWS=$mode b --workload $wl --layout shell; done; done 2>&1 | tee -a $O/ab.txt
