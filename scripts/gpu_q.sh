#!/bin/bash
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for wl in c4 c3 c2; do
  for b in 0 1; do
    echo "== $wl BARRIER=$b"; GDR_FWD_BARRIER=$b run --workload $wl; GDR_FWD_BARRIER=$b run --workload $wl --layout shell
  done
done
