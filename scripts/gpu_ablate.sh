#!/bin/bash
# serial (GDR_RENDER_SIDE=0) per-kernel durations of the ablation builds at C4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 0 ${KS:-1 2 3 4 5 6}; do
  if [ $k = 0 ]; then unset GDR_LIB_PATH; else export GDR_LIB_PATH=$PWD/build/libgdr_abl$k.so; fi
  GDR_RENDER_SIDE=0 python bench.py --workload ${WL:-c4} --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python scripts/_kt.py abl$k
done
