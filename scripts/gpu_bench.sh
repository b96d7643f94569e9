#!/bin/bash
# GPU box: bench (c4 + c2) and a rocprofv3 kernel-trace of a short c4 run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "rc=$?" >> gpurun_out/bench_c4.err
cat gpurun_out/bench_c4.json; tail -5 gpurun_out/bench_c4.err
timeout 600 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "rc=$?" >> gpurun_out/bench_c2.err
cat gpurun_out/bench_c2.json; tail -3 gpurun_out/bench_c2.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_c4 -o c4 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/gpurun_out/prof_c4.log 2>&1)
ls -R gpurun_out/prof_c4 | head -20
