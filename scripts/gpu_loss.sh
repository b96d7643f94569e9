#!/bin/bash
# fused-loss A/B: tests, then c4/c2/c3 with the fused HIP loss (default) and with torch ops
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for wl in c4 c2 c3; do for mode in "" "--torch-loss"; do
  timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline $mode > gpurun_out/bench_$wl.json 2>gpurun_out/bench_$wl.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$wl.json')); print('$wl [$mode]', d['value'], 'views/s', d['ms_per_step'], 'ms/step', 'loss', d['loss_mean'], 'path_frac', d['roofline']['path_frac']); print('  ', {k:v['avg_us'] for k,v in d['kernels'].items()})" || tail -5 gpurun_out/bench_$wl.err
done; done
