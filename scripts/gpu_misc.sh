#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -iE "RDREQ|WRREQ|FETCH_SIZE|WRITE_SIZE|EA0_RD|MALL" | head -40) > gpurun_out/counters.txt 2>&1
head -40 gpurun_out/counters.txt
echo "--- torchrun 2 ranks on one GPU (gloo), c2"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 --dist-backend gloo --single-device --no-roofline 2>&1 | tail -3
echo "--- torchrun 2 ranks on one GPU (nccl=RCCL), c2"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 --single-device --no-roofline 2>&1 | tail -3
echo "--- default bench"
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
