#!/bin/bash
# per-kernel launch durations with every kernel on ONE stream (GDR_RENDER_SIDE=0): what each kernel costs alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for wl in ${WLS:-c4 c3 c2}; do for mode in "" "--per-view --unfused"; do
  tag=$wl$( [ -n "$mode" ] && echo _pv )
  GDR_RENDER_SIDE=0 timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline $mode > gpurun_out/serial_$tag.json 2>gpurun_out/serial_$tag.err
  python - gpurun_out/serial_$tag.json "$wl [$mode] serial" <<'P' || tail -5 gpurun_out/serial_$tag.err
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'],'views/s', d['ms_per_step'],'ms/step')
tot=sum(v['total_ms'] for v in d['kernels'].values())/d['steps']
print('   kernel ms/step total', round(tot,3))
for k,v in d['kernels'].items(): print('   %-22s avg %8.2f us x %5.1f /step = %7.3f ms'%(k,v['avg_us'],v['launches']/d['steps'],v['total_ms']/d['steps']))
P
done; done
