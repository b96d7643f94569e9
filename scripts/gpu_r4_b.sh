#!/bin/bash
# round 4, run B: native per-view forward + surfel render groups — the whole GPU suite, host split, per-view bench legs;
# 2DGS numerics A/B (GSR_PRECISE on = in-tree lib, off = variants/libgdr_surfel_fast.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python scripts/host_split.py 2>/dev/null | tail -1 | tee $O/host_split.txt
python scripts/host_split2.py 2>/dev/null | grep "us per call" | head -16 >> $O/host_split.txt
b() { python bench.py "$@" --steps 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'per_view', (d.get('per_view') or {}).get('value'))"; }
for wl in c4 c2 c3 c5; do echo -n "$wl: "; b --workload $wl; done 2>&1 | tee $O/bench.txt
echo -n "c3step: "; python bench.py --workload c3step --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'per_view', (d.get('per_view') or {}).get('value'))" | tee -a $O/bench.txt
echo "--- 2DGS numerics: precise (in-tree) vs fast (round-3 arithmetic)"
for lib in "" "$PWD/generativedensification_amd/lib/variants/libgdr_surfel_fast.so"; do
  GDR_LIB_PATH=$lib timeout 1200 python -m pytest tests/test_gpu_oracle_fullsize.py -q -rP -k "surfel" 2>&1 | grep -E "^\[|passed|failed" | cut -c1-260 | tee -a $O/surfel_numerics.txt
  echo -n "c5 bench lib=${lib:-intree}: "; GDR_LIB_PATH=$lib b --workload c5 --no-per-view-leg | tee -a $O/surfel_numerics.txt
done
