#!/bin/bash
# End-of-round evidence run on the GPU box: smoke + tests (with the parity statistics), bench lines of every workload
# (each carries its `per_view` second headline), the reference train-step sequence, rocprofv3 kernel stats, PMC traffic
# (-> profiles/pmc_traffic.json), a kernel timeline, the scene-size sweep, 2- and 8-rank smoke.  Everything lands in
# gpurun_out/final/ (copied into profiles/ afterwards).   bash scripts/gpu_final.sh r03
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
TAG=${1:-r06}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log
grep -E "^\[" $O/pytest_gpu.log > $O/${TAG}_fullsize_parity.log
echo "--- float-atomic rate of the device (scripts/ubench/atomic_probe.hip): what bounds K7"
P=generativedensification_amd/lib/atomic_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ubench/atomic_probe.hip -o $P 2>/dev/null
timeout 120 $P | tee $O/${TAG}_atomic_probe.txt
echo "--- PMC passes first: the bench lines below read the traffic table they produce"
for wl in c4 c3 c2 c5; do
  BENCH_ARGS="--workload $wl --no-per-view-leg" bash scripts/gpu_pmc.sh pmc_$wl > $O/pmc_$wl.log 2>&1
  cp gpurun_out/pmc_${wl}_summary.json $O/${TAG}_${wl}_pmc_summary.json 2>/dev/null; rm -rf gpurun_out/pmc_${wl}_[0-9]*
  tail -4 $O/pmc_$wl.log | cut -c1-300
done
python scripts/make_pmc_traffic.py $O/${TAG} && cp $O/pmc_traffic.json profiles/pmc_traffic.json
echo "--- K7 stall breakdown at the reference's own scale (pairs kernel pinned), round-4 verdict next #4"
for wl in c2 c3 c4; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-per-view-leg > $O/k7line_$wl.json 2>/dev/null
  python scripts/k7_stalls.py $O/${TAG}_${wl}_pmc_summary.json $O/k7line_$wl.json $O/${TAG}_k7_stalls_$wl.json | cut -c1-400
done
python scripts/binning_stalls.py $O/${TAG}_c4_pmc_summary.json $O/k7line_c4.json $O/${TAG}_binning_stalls.json | cut -c1-600
echo "--- K7 / K6 phase budget from measurement builds (csrc/render.hip GDR_K7_STUB; built here if absent)"
for st in 1 2 4 8 16 32 3 7 23; do [ -f generativedensification_amd/lib/variants/libgdr_hip_k7stub$st.so ] || make -C generativedensification_amd/csrc variant VTAG=k7stub$st VDEFS=-DGDR_K7_STUB=$st > /dev/null 2>&1; done
bash scripts/gpu_k7_budget.sh > $O/k7_budget.log 2>&1; cp gpurun_out/k7_budget/k7_budget.json $O/${TAG}_k7_budget.json 2>/dev/null; tail -3 $O/k7_budget.log | cut -c1-300
for u in valu_rate valu_select; do P=generativedensification_amd/lib/$u; [ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/$u.hip -o $P 2>/dev/null; timeout 120 $P > $O/${TAG}_$u.txt 2>&1; done
for k in 20 21 22 23 24 25 26 27; do timeout 15 generativedensification_amd/lib/valu_select $k >> $O/${TAG}_valu_select.txt 2>&1; done
b() { name=$1; shift; timeout 1200 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_bench_$name.json")); r = d["roofline"] or {}; pv = d.get("per_view") or {}
    print("$name", d["value"], d["unit"], d["ms_per_step"], "ms/step | per_view", pv.get("value"), "| D", d["config"].get("num_rendered_per_view"), "dom", r.get("kernel"), r.get("frac"), r.get("frac_serial"), "atomic", r.get("atomic_frac_serial"), "path", r.get("path_frac"), r.get("path_frac_measured"), "cpu", (d["cpu_baseline"] or {}).get("value"))
except Exception as e: print("$name FAILED", e)
PY
}
b default
b c2 --workload c2
b c3 --workload c3
b c5 --workload c5
b c3step --workload c3step --steps 6 --warmup 2
b c5_imageloss --workload c5 --image-loss --no-cpu-baseline
b c5_imageloss_perview --workload c5 --image-loss --per-view --unfused --no-cpu-baseline
b c4_fwd --forward-only
b c2_fwd --workload c2 --forward-only
b c3_fwd --workload c3 --forward-only
b c5_fwd --workload c5 --forward-only
b c4_backward_per_view --backward-per-view --unfused --no-cpu-baseline
b c4_perview --per-view --unfused --no-cpu-baseline
b c4_imagesout --torch-loss --no-cpu-baseline --no-per-view-leg
b c4_shell --layout shell --no-cpu-baseline
b c2_shell --workload c2 --layout shell --no-cpu-baseline
b c3_shell --workload c3 --layout shell --no-cpu-baseline
b c5_shell --workload c5 --layout shell --no-cpu-baseline
b c3step_shell --workload c3step --layout shell --steps 6 --warmup 2
GDR_GROUP_VIEWS=0 timeout 600 python bench.py --per-view --unfused --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c4 per-view, render groups OFF', d['value'])"
for wl in c4 c3 c2 c5; do
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 8 --warmup 12 --no-cpu-baseline --no-roofline --no-per-view-leg > $O/prof_$wl.log 2>&1)   # (20 steps: launches 8-11 of a shape time the two K7 kernels, the choice serves from launch 12)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${wl}_kernel_stats.csv && python scripts/stats_print.py $f 20 8
  rm -rf $O/prof_$wl
done
echo "--- kernel timeline of a C4 step + idle gaps"
bash scripts/gpu_timeline.sh c4 --no-per-view-leg > /dev/null 2>&1; cp gpurun_out/timeline_c4.txt $O/${TAG}_timeline_c4.txt; head -12 $O/${TAG}_timeline_c4.txt | grep "^step"
echo "--- the unchanged caller's loop: kernel timeline + GPU-busy fraction (C4 GPU-bound, C2 80 % busy)"
for wl in c4 c2 c5; do bash scripts/gpu_timeline_pv.sh $wl > /dev/null 2>&1; cp gpurun_out/timeline_pv_$wl.txt $O/${TAG}_timeline_pv_$wl.txt; grep "^step [23]" $O/${TAG}_timeline_pv_$wl.txt; done
echo "--- host time of the unchanged caller's loop: compiled boundary (csrc/boundary.cpp) vs the python + ctypes boundary"
python scripts/host_split.py 2>/dev/null | grep -E "host boundary|^N=" | tee $O/${TAG}_host_split.txt
GDR_COMPILED_BOUNDARY=0 python scripts/host_split.py 2>/dev/null | grep -E "host boundary|^N=" | tee -a $O/${TAG}_host_split.txt
GDR_COMPILED_BOUNDARY=0 python scripts/host_split2.py 2>/dev/null | grep "us per call" | head -16 >> $O/${TAG}_host_split.txt
echo "--- same box A/B: compiled boundary on / off, forward reuse on / off (unchanged caller, cameras built once)" | tee $O/${TAG}_ab_boundary_reuse.txt
for wl in c5 c2 c3; do for cb in 1 0; do for rf in 1 0; do
  GDR_COMPILED_BOUNDARY=$cb GDR_REUSE_FORWARD=$rf timeout 600 python bench.py --workload $wl --per-view --unfused --prebuilt-cams --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl compiled_boundary=$cb reuse_forward=$rf', d['value'], 'views/s', d['ms_per_step'], 'ms/step')" | tee -a $O/${TAG}_ab_boundary_reuse.txt
done; done; done
echo "--- abs-grad entry + device top-k"
python scripts/absgrad_bench.py 2>/dev/null | tee $O/${TAG}_absgrad.txt
echo "--- size sweep (every frac must stay <= 1; traffic null off the recorded scene)"
for n in 500000 2000000 8000000 32000000; do
  timeout 900 python bench.py --n $n --steps 20 --warmup 3 --no-cpu-baseline --no-per-view-leg > $O/n$n.json 2> $O/n$n.err || { echo "N=$n FAILED"; tail -3 $O/n$n.err; continue; }
  python -c "
import json; d=json.load(open('$O/n$n.json')); r=d['roofline'] or {}
fr=[k.get('frac',0) for k in d['kernels'].values()]+[k.get('frac_serial',0) or 0 for k in d['kernels'].values()]
print(json.dumps(dict(n=$n, views_per_s=d['value'], ms_per_step=d['ms_per_step'], D=d['config']['num_rendered_per_view'], path_frac=r.get('path_frac'), path_frac_measured=r.get('path_frac_measured'), traffic=r.get('traffic'), max_kernel_frac=max(fr), mem_gb=d['config'].get('peak_mem_gb'))))" | tee -a $O/${TAG}_size_sweep.json
done
echo "--- RCCL, one rank, collectives forced: plain / K9 writing into the packed buffer (gradient sinks) / the same with the collectives asynchronous / gradients kept"
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-per-view-leg 2>/dev/null | tail -1 | tee $O/${TAG}_bench_rccl_1rank_plain.json | cut -c1-120
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce 2>/dev/null | tail -1 | tee $O/${TAG}_bench_rccl_1rank.json | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce --separate-loss-gather 2>/dev/null | tail -1 | tee $O/${TAG}_bench_rccl_1rank_separate_loss_gather.json | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce --overlap-comm 2>/dev/null | tail -1 | tee $O/${TAG}_bench_rccl_1rank_overlap.json | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-per-view-leg --force-dist --grad-allreduce --keep-grads 2>/dev/null | tail -1 | tee $O/${TAG}_bench_rccl_1rank_keepgrads.json | cut -c1-200
echo "--- the reference's per-sample sequence through the unchanged caller: phases, forward reuse on / off, kernel timeline"
python scripts/c3step_phases.py 2>/dev/null | tee $O/${TAG}_c3step_phases.txt | grep -E "^---|sum|unsynchronised"
python scripts/c3step_phases.py --fresh-cams 2>/dev/null | tee -a $O/${TAG}_c3step_phases.txt | grep -E "^---|sum|unsynchronised"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_c3step -o t -- python $R/scripts/c3step_phases.py --trace > $R/gpurun_out/trace_c3step.log 2>&1)
f=$(find gpurun_out/trace_c3step -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_gaps.py $f --every 16 --timeline > $O/${TAG}_timeline_c3step.txt; rm -rf gpurun_out/trace_c3step; head -3 $O/${TAG}_timeline_c3step.txt
echo "--- 2DGS parity statistics at other absolute floors (tests/util.py assert_grads_surfel)"
GDR_TEST_STATS=1 timeout 1200 python -m pytest tests/test_gpu_oracle_fullsize.py tests/test_gpu_surfel.py -q -rP -k "surfel" 2>&1 | grep -E "^\[|passed|failed" | cut -c1-200 > $O/${TAG}_surfel_stats.txt; tail -1 $O/${TAG}_surfel_stats.txt
bash scripts/gpu_timeline.sh c2 --no-per-view-leg > /dev/null 2>&1; cp gpurun_out/timeline_c2.txt $O/${TAG}_timeline_c2.txt; grep "^step [45]" $O/${TAG}_timeline_c2.txt
echo "--- 2 and 8 ranks on one GPU (gloo)"
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 --single-device --no-roofline --no-per-view-leg 2>/dev/null | tail -1 | tee $O/${TAG}_bench_2ranks_c2.json | cut -c1-260
timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --workload c2 --single-device --no-roofline --no-per-view-leg --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/${TAG}_bench_8ranks_c2.json | cut -c1-360
