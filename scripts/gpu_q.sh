#!/bin/bash
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'], o['loss_mean'])"; }
for rep in 1 2; do for wl in c4 c3; do
  echo "== $wl base"; run --workload $wl
  echo "== $wl no memset (upper bound; gradients wrong)"; GDR_EXP_NOMEMSET=1 run --workload $wl
  echo "== $wl no memset + K9 zeroes the records it reads"; GDR_EXP_NOMEMSET=1 GDR_EXP_K9ZERO=1 run --workload $wl
done; done
