#!/bin/bash
# Size sweep of the C4 scene (N Gaussians, 4 views of 800x800) + a one-rank RCCL pass through the N>1 code path.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/scale; mkdir -p $O
echo "--- RCCL, one rank, collectives forced (barrier, loss all-gather, grad reduce-scatter + all-gather)"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --force-dist --grad-allreduce > $O/rccl1.json 2> $O/rccl1.err || tail -5 $O/rccl1.err
python -c "
import json; d=json.load(open('$O/rccl1.json')); print('rccl world1', d['value'], d['ms_per_step'], d['loss_mean'])"
for n in 500000 1000000 2000000 4000000 8000000 16000000 32000000; do
  timeout 900 python bench.py --n $n --steps 4 --warmup 2 --no-cpu-baseline > $O/n$n.json 2> $O/n$n.err || { echo "N=$n FAILED"; tail -3 $O/n$n.err; continue; }
  python -c "
import json; d=json.load(open('$O/n$n.json')); r=d['roofline'] or {}
print('N=$n', 'views/s', d['value'], 'ms/step', d['ms_per_step'], 'D', d['config']['num_rendered_per_view'], 'path_frac', r.get('path_frac'), 'loss', d['loss_mean'], 'mem GB', d['config'].get('peak_mem_gb'))"
done
