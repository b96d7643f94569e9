#!/bin/bash
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_surfel.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
for wl in c4 c3 c2 c5; do
  for b in 8 2 1; do
    echo "== $wl BWD_GROUP=$b"; GDR_BWD_GROUP=$b run --workload $wl; GDR_BWD_GROUP=$b run --workload $wl --layout shell
  done
done
