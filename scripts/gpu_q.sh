#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_surfel.py -x -q 2>&1 | tail -2
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for wl in c4 c2 c5 c3; do for lay in cube shell; do echo "== $wl $lay"; run --workload $wl --layout $lay; run --workload $wl --layout $lay; done; done
