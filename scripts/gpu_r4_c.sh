#!/bin/bash
# round 4, run C: whole GPU suite (all failures), surfel oracle-order numerics A/B, host split, unchanged-caller timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
echo "--- 2DGS numerics: oracle order (in-tree) vs round-3 arithmetic"
for lib in "" "$PWD/generativedensification_amd/lib/variants/libgdr_surfel_fast.so"; do
  GDR_LIB_PATH=$lib timeout 1200 python -m pytest tests/test_gpu_oracle_fullsize.py -q -rP -k "surfel" 2>&1 | grep -E "^\[|passed|failed" | cut -c1-260 >> $O/surfel_numerics.txt
  echo "c5 bench lib=${lib:-intree}: $(GDR_LIB_PATH=$lib python bench.py --workload c5 --steps 10 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])")" | tee -a $O/surfel_numerics.txt
done
grep -E "render_views c5|whole image\]|^\[c5\]|bench" $O/surfel_numerics.txt | cut -c1-200
python scripts/host_split.py 2>/dev/null | tail -1 | tee $O/host_split.txt
python scripts/host_split2.py 2>/dev/null | grep "us per call" | head -12 >> $O/host_split.txt
for wl in c2 c5; do bash scripts/gpu_timeline_pv.sh $wl > /dev/null 2>&1; cp gpurun_out/timeline_pv_$wl.txt $O/; grep "^step [23]" $O/timeline_pv_$wl.txt; done
