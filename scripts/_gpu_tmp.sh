#!/bin/bash
cd /root/repo
run() {
  for w in "c4" "c2" "c3" "c2 --layout shell" "c3 --layout shell" "c4 --layout shell"; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-per-view-leg 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=d['kernels']['render_bwd']
        print('$TAG', '$w', round(d['value'],1), 'K7 us', k['avg_us'], 'serial', r.get('avg_launch_us_serial') if r.get('kernel')=='render_bwd' else None)
"
  done
}
for rep in 1 2; do
export TAG=old GDR_LIB_PATH=$PWD/generativedensification_amd/lib/old/libgdr_hip.so; run
unset GDR_LIB_PATH
export TAG=new_off GDR_PAIR_W=1000000; run
export TAG=new_w6 GDR_PAIR_W=6; run
done
