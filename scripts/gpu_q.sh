#!/bin/bash
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"; }
for wl in c3 c4 c2; do
  echo "== $wl base"; run --workload $wl
  for q in 8 16; do for f in 4 8; do
    echo "== $wl GPU_MAX_HW_QUEUES=$q FWD_STREAMS=$f"; GPU_MAX_HW_QUEUES=$q GDR_FWD_STREAMS=$f run --workload $wl
  done; done
  echo "== $wl HWQ=8 FWD=8 BWD_STREAMS=4"; GPU_MAX_HW_QUEUES=8 GDR_FWD_STREAMS=8 GDR_BWD_STREAMS=4 run --workload $wl
done
