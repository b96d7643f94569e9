"""profiles/pmc_traffic.json from the per-workload PMC summaries of scripts/gpu_pmc.sh:
python scripts/make_pmc_traffic.py profiles/r03   ->  {wl: {kernel: HBM bytes per launch}, wl_valu: {kernel: SQ_INSTS_VALU}, wl_atomic: {kernel: TCC_EA0_ATOMIC},
wl_meta: {n, layout, order, views_per_gpu, abi}} — bench.py uses a table only for the scene recorded in its `_meta` (the PMC
passes run `bench.py --workload wl` with its default scene) and prints `traffic: null` for any other.
(bytes per launch = `_traffic_bytes_per_launch` of the summary: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB, the gfx950 reading
of MI355X_MICROARCH.md's HBM section, see profiles/README.md)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from generativedensification_amd._lib import ABI_VERSION
prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
out = {}
for wl in ("c4", "c3", "c2", "c5"):
    f = f"{prefix}_{wl}_pmc_summary.json"
    if not os.path.exists(f):
        continue
    s = json.load(open(f))
    out[wl] = {k: int(v) for k, v in s["_traffic_bytes_per_launch"].items()}
    out[wl + "_valu"] = {k: int(v["SQ_INSTS_VALU"]) for k, v in s.items() if not k.startswith("_") and "SQ_INSTS_VALU" in v}
    # float-atomic record lines that left the L2s (K7's publish: one per (entry, block) hit) — the rate of these, not HBM
    # bytes or VALU issue, is what bounds K7 (scripts/atomic_probe.hip: ~21 G lines / s for the device)
    out[wl + "_atomic"] = {k: int(v["TCC_EA0_ATOMIC_sum"]) for k, v in s.items()
                           if not k.startswith("_") and v.get("TCC_EA0_ATOMIC_sum")}
    out[wl + "_meta"] = dict(n=WORKLOADS[wl]["n"], layout="cube", order="random", views_per_gpu=WORKLOADS[wl]["views_per_gpu"],
                             abi=ABI_VERSION,     # (the launch shapes belong to a library version: K7 is one launch per node since v14)
                             source=os.path.basename(f))
json.dump(out, open(os.path.join(os.path.dirname(prefix), "pmc_traffic.json"), "w"))
print({k: len(v) for k, v in out.items()})
