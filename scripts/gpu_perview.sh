#!/bin/bash
# the unchanged caller's pattern (render_img per view, one backward): views/s with and without render groups, per workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for wl in ${WLS:-c4 c3 c2 c5}; do for grp in 1 0; do
  GDR_GROUP_VIEWS=$grp timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --per-view --unfused ${EXTRA} > gpurun_out/pv_${wl}_g$grp.json 2>gpurun_out/pv_${wl}_g$grp.err
  python - gpurun_out/pv_${wl}_g$grp.json "$wl per-view unfused groups=$grp" <<'P' || tail -5 gpurun_out/pv_${wl}_g$grp.err
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'],'views/s', d['ms_per_step'],'ms/step', 'kernel ms/step', round(sum(v['total_ms'] for v in d['kernels'].values())/d['steps'],3))
print('    ', {k:(v['avg_us'], round(v['launches']/d['steps'],1)) for k,v in d['kernels'].items()})
P
done; done
