#!/bin/bash
# tests + short bench lines (c4, c2) with per-kernel averages
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for wl in c4 c2; do for mode in "" "--per-view --unfused" "--backward-per-view --unfused"; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline $mode > gpurun_out/bench_$wl.json 2>gpurun_out/bench_$wl.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$wl.json')); print('$wl [$mode]', d['value'], 'views/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'], d['roofline']['frac'], 'path_frac', d['roofline']['path_frac']); print('  ', {k:v['avg_us'] for k,v in d['kernels'].items()})" || tail -5 gpurun_out/bench_$wl.err
done; done
