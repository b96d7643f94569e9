#!/bin/bash
# 2DGS surfel workload: bench line with per-kernel averages (+ unfused variant), optional rocprof kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for mode in "" "--unfused"; do
  timeout 900 python bench.py --workload c5 --steps 6 --warmup 2 $mode ${C5_EXTRA:---no-cpu-baseline} > gpurun_out/bench_c5.json 2>gpurun_out/bench_c5.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_c5.json')); print('c5 [$mode]', d['value'], 'views/s', d['ms_per_step'], 'ms/step', 'D', d['config']['num_rendered_per_view'], d['roofline']['kernel'], d['roofline']['frac'], 'path_frac', d['roofline']['path_frac'], 'loss', d['loss_mean']); print('  ', {k:v['avg_us'] for k,v in d['kernels'].items()}); print('  cpu', d['cpu_baseline'])" || tail -5 gpurun_out/bench_c5.err
done
if [ -n "$C5_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_c5" -o c5 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_c5.log" 2>&1
  cd "$GRAFT_REPO_ROOT"; python scripts/stats_print.py $(find gpurun_out/prof_c5 -name '*kernel_stats.csv' | head -1) 22
fi
