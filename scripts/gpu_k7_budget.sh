#!/bin/bash
# GPU box: the K7 (and K6 cull) instruction / time budget asked for by the round-5 verdict (next #1a).
# Measurement builds of render.hip with ONE phase of the walk compiled out (or twice in) — csrc/render.hip GDR_K7_STUB,
# built by `make variant` into generativedensification_amd/lib/variants/ — are run on the same box, same process layout:
#   * launch time of K7 (row kernel pinned, GDR_K7_PAIRS=0) and of K6 from bench.py's serial per-kernel events,
#   * SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAVE_CYCLES per launch from a separate rocprofv3 --pmc pass (--kernel-trace only).
# Output: gpurun_out/k7_budget/{time,pmc}_<variant>_<workload>.json -> scripts/k7_budget.py -> profiles/r06_k7_budget.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/k7_budget; mkdir -p $O
VARIANTS=${VARIANTS:-"release k7stub1 k7stub2 k7stub4 k7stub8 k7stub16 k7stub32 k7stub3 k7stub7 k7stub23"}
WORKLOADS=${WORKLOADS:-"c4 c3 c2"}
PMC_WORKLOADS=${PMC_WORKLOADS:-"c4 c3"}
export GDR_ALLOW_EXPERIMENTAL_LIB=1 GDR_K7_PAIRS=0
ARGS="--steps 10 --warmup 4 --no-settle --no-cpu-baseline --no-per-view-leg"
for v in $VARIANTS; do
  if [ $v = release ]; then unset GDR_LIB_PATH; else export GDR_LIB_PATH=$R/generativedensification_amd/lib/variants/libgdr_hip_$v.so; [ -f $GDR_LIB_PATH ] || { echo "missing $GDR_LIB_PATH"; continue; }; fi
  for wl in $WORKLOADS; do
    timeout 300 python bench.py --workload $wl $ARGS > $O/time_${v}_$wl.json 2> $O/time_${v}_$wl.err || tail -2 $O/time_${v}_$wl.err
  done
  for wl in $PMC_WORKLOADS; do
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
        --kernel-trace --output-format csv -d $O/pmc_${v}_$wl -o p -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --no-settle --no-cpu-baseline --no-roofline --no-per-view-leg > $O/pmc_${v}_$wl.log 2>&1)
    python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/pmc_${v}_$wl/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gdr::" not in k or "render_" not in k: continue
        name = k.split("gdr::(anonymous namespace)::")[1].split("(")[0].split("<")[0]
        out[name][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[name][row["Counter_Name"]] += 1
res = {k: {c: out[k][c] / max(cnt[k][c], 1) for c in out[k]} for k in out}
res["_launches"] = {k: max(cnt[k].values()) for k in cnt}
json.dump(res, open("$O/pmc_${v}_$wl.json", "w"), indent=1)
PY
    rm -rf $O/pmc_${v}_$wl
  done
  echo "$v done"
done
python scripts/k7_budget.py $O $O/k7_budget.json | tail -40
