#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_surfel.py tests/test_gpu_viewgroup.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_oracle_fullsize.py -x -q -m gpu -k surfel 2>&1 | tail -3
run() {
  for a in "" "--image-loss" "--per-view --unfused --image-loss" "--layout shell" "--layout shell --image-loss"; do
    timeout 300 python bench.py --workload c5 $a --steps 20 --warmup 5 --no-cpu-baseline --no-per-view-leg 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels'].get('render_bwd',{})
        print('$TAG', 'c5 $a', round(d['value'],1), d['ms_per_step'], 'K7s us', k.get('avg_us'), 'serial', k.get('avg_us_serial'), 'launches', k.get('launches'))
"
  done
}
for rep in 1 2; do
export TAG=old GDR_LIB_PATH=$PWD/generativedensification_amd/lib/old/libgdr_hip.so; run
unset GDR_LIB_PATH; export TAG=new; run
done
