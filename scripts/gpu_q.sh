#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_boundary.py -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for h in 0 1 0 1; do echo "== c4 per-view unfused HINTS=$h"; GDR_LAUNCH_HINTS=$h run --per-view --unfused; done
