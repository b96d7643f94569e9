#!/bin/bash
# A/B of an env-controlled knob on one box: VAR=name VALUES="a b c" WL=c5 bash scripts/gpu_ab.sh  (2 rounds each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for round in 1 2; do for v in $VALUES; do
  env $VAR=$v timeout 600 python bench.py --workload ${WL:-c5} --steps 8 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernels']; print('$VAR=$v', d['value'], 'views/s', 'render_fwd', k['render_fwd']['avg_us'], 'render_bwd', k['render_bwd']['avg_us'])" || tail -3 gpurun_out/ab.err
done; done
