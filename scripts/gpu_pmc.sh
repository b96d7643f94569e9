#!/bin/bash
# GPU box: rocprofv3 PMC passes (counters only + kernel-trace) on a short bench run; summarised per kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-pmc}
R=$PWD
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_ARGS}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$i -o p -- python $R/bench.py $ARGS > $R/gpurun_out/${TAG}_$i.log 2>&1)
done
# the row-pair K7 (render_bwd_pairs_kernel: the library picks it per scene shape after timing both, which a two-step run never
# reaches): its instruction, traffic and atomic-line counts from runs that pin it
# (round 5: also its wave-cycle / stall counters — the per-wave stall breakdown of profiles/rNN_k7_stalls_*.json)
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp GDR_K7_PAIRS=1 && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$i -o p -- python $R/bench.py $ARGS > $R/gpurun_out/${TAG}_$i.log 2>&1)
done
python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/${TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gdr::" not in k: continue
        name = k.split("gdr::(anonymous namespace)::")[1].split("(")[0].split("<")[0]
        c = row["Counter_Name"]; v = float(row["Counter_Value"])
        out[name][c] += v; cnt[name][c] += 1
res = {k: {c: out[k][c] / max(cnt[k][c], 1) for c in out[k]} for k in out}
res["_launches"] = {k: max(cnt[k].values()) for k in cnt}
# HBM traffic per launch (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE
# reports half of the bytes read (calibrated here on preprocess_fwd / tile_ranges / sort_hist, whose read
# byte counts are known: 2x matches them), WRITE_SIZE is accurate (calibrated on preprocess_fwd).
res["_traffic_bytes_per_launch"] = {k: int(2 * 1024 * res[k].get("FETCH_SIZE", 0) + 1024 * res[k].get("WRITE_SIZE", 0))
                                    for k in res if not k.startswith("_")}
json.dump(res, open("gpurun_out/${TAG}_summary.json", "w"), indent=1)
for k in res:
    if k.startswith("_"): continue
    print(k, {c: (round(v) if v > 100 else round(v, 3)) for c, v in sorted(res[k].items())})
PY
