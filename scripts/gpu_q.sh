#!/bin/bash
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for wl in c4 c3 c2; do
  echo "== $wl full"; run --workload $wl
  for m in 0 4 6 3; do echo "== $wl ABL_TSORT=$m (bit0 long, bit1 medium, bit2 small launched)"; GDR_ABL_TSORT=$m run --workload $wl; done
done
