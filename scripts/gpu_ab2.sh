#!/bin/bash
# A/B with per-kernel durations: VAR=name VALS="a b" WLS="c4 c2" LAYOUTS="cube shell" bash scripts/gpu_ab2.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for wl in ${WLS:-c4 c2 c3}; do for lay in ${LAYOUTS:-cube}; do for val in ${VALS:-0 1}; do
  env ${VAR:-GDR_GROUP_LANES}=$val python bench.py --workload $wl --layout $lay --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline ${EXTRA} 2>/dev/null | python scripts/_kt.py "$wl-$lay-${VAR:-GDR_GROUP_LANES}=$val"
done; done; done
