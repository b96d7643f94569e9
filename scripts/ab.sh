#!/bin/bash
# A/B of render variants in separate processes (GDR_RENDER_VARIANT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in ${VARIANTS:-3 4}; do for wl in c4 c2; do
  GDR_RENDER_VARIANT=$v python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v $wl', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'render' in k})"
done; done
