#!/bin/bash
# round 4, run F: whole GPU suite; fused backward with K7 launched first (same-box A/B against the previous commit is not
# possible: numbers against run E's); -fno-slp-vectorize build of the render kernels A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"; }
NOSLP=$PWD/generativedensification_amd/lib/variants/libgdr_noslp.so
for wl in c4 c3 c2 c5; do for rep in 1 2; do
  echo -n "$wl default rep$rep: "; b --workload $wl
  echo -n "$wl noslp   rep$rep: "; GDR_LIB_PATH=$NOSLP b --workload $wl
done; done 2>&1 | tee $O/ab.txt
