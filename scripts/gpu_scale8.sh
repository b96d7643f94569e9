#!/bin/bash
# The multi-GPU scaling run, ready for the day an 8-GPU MI355X node is available (no N > 1 RCCL run on hardware exists yet:
# the driver's 8-GPU tier was never available, SCALE_r01..r04 are "skipped").  One rank per GPU over RCCL / xGMI:
#   bash scripts/gpu_scale8.sh [out_dir]
# writes out_dir/scale_<workload>_<N>.json for N = 1 2 4 8 (render-only scaling: loss all-gather) and the same with the
# gradient reduce-scatter + all-gather (--grad-allreduce, plain and --overlap-comm); comm_ms.ranks_seen in every line
# says how many ranks the collectives really spanned, comm_ms.expected_grad_allreduce_ms what xGMI allows.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/scale8}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for wl in c4 c2; do for n in 1 2 4 8; do
  [ $n -le $NG ] || continue
  for mode in "" "--grad-allreduce" "--grad-allreduce --overlap-comm"; do
    tag=$(echo "$mode" | tr -d ' -'); tag=${tag:-render}
    if [ $n -eq 1 ]; then cmd="python bench.py"; else
      cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py"; fi
    timeout 900 $cmd --gpus $n --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-per-view-leg $mode \
      > $OUT/scale_${wl}_${n}_${tag}.json 2> $OUT/scale_${wl}_${n}_${tag}.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/scale_${wl}_${n}_${tag}.json"))
    print("$wl N=$n $tag:", d["value"], "views/s", d["ms_per_step"], "ms/step", "ranks", (d.get("comm_ms") or {}).get("ranks_seen"),
          "comm", {k: v for k, v in (d.get("comm_ms") or {}).items() if k in ("loss_gather", "grad_allreduce")})
except Exception as ex:
    print("$wl N=$n $tag: FAILED", ex)
PY
  done
done; done
