#!/bin/bash
# A/B of one environment switch over the bench workloads: VAR=name VALS="0 1" bash scripts/gpu_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for wl in ${WLS:-c4 c2 c3 c5}; do for lay in ${LAYOUTS:-cube shell}; do for val in ${VALS:-0 1}; do
  env ${VAR:-GDR_RENDER_SIDE}=$val python bench.py --workload $wl --layout $lay --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-roofline ${EXTRA} > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$wl $lay ${VAR:-GDR_RENDER_SIDE}=$val', d['value'], d['ms_per_step'], d['loss_mean'])"
done; done; done
