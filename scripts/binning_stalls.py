#!/usr/bin/env python
"""What the binning chain's kernels are waiting on (round-5 verdict next #7): per kernel of the chain — tile_count, tile_scan,
tile_order, tile_scatter, tile_sort — the launch duration (bench.py's serial per-kernel events), the waves it puts on the chip,
the mean resident waves per SIMD over its launch, the shares of its wave cycles spent issuing / parked on s_waitcnt or a barrier
/ issue-stalled, and its HBM traffic against the launch time.  From a scripts/gpu_pmc.sh summary + a bench line.

    python scripts/binning_stalls.py gpurun_out/pmc_c4_summary.json bench_line.json out.json
"""
import json
import sys

pmc, bench, out = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
SIMDS, CLK = 1024, 2.2e9
names = {"tile_count_kernel": "tile_count", "tile_scan_kernel": "tile_scan", "tile_order_kernel": "tile_order",
         "tile_scatter_kernel": "tile_scatter", "tile_sort_kernel": "tile_sort"}
traffic = pmc.get("_traffic_bytes_per_launch", {})
res = {"workload": bench["config"]["workload"], "clock_hz_assumed": CLK, "kernels": {},
       "note": "us = launch duration alone on one stream (HIP events); resident waves per SIMD = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x "
               "launch cycles); the shares are of the kernel's summed wave cycles (quad-cycle counters); tile_sort's counters are the "
               "mean over its size-class launches (three per view), its us the sum of them"}
chain = 0.0
for kn, bn in names.items():
    c = pmc.get(kn)
    if not c or "SQ_WAVE_CYCLES" not in c:
        continue
    kb = bench["kernels"].get(bn, {})
    us = kb.get("avg_us_serial") or kb.get("avg_us") or 0.0
    if bn == "tile_sort":
        kl = bench["kernels"].get("tile_sort_long", {})
        per_view = max(1, round(kl.get("launches", 0) / max(kb.get("launches", 1), 1)))
        us = us + per_view * (kl.get("avg_us_serial") or kl.get("avg_us") or 0.0)
    wc = c["SQ_WAVE_CYCLES"]
    d = dict(us=round(us, 1), waves=int(c.get("SQ_WAVES", 0)),
             resident_waves_per_simd=round(wc * 4.0 / (SIMDS * max(us, 1e-3) * 1e-6 * CLK), 2),
             share=dict(issuing=round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), parked_waitcnt_or_barrier=round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                        issue_stalled=round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)),
             valu=int(c.get("SQ_INSTS_VALU", 0)), lds=int(c.get("SQ_INSTS_LDS", 0)),
             hbm_bytes=int(traffic.get(kn, 0)), hbm_GBs=round(traffic.get(kn, 0) / max(us, 1e-3) / 1e3, 1) if traffic.get(kn) else None)
    res["kernels"][bn] = d
    chain += us
res["chain_us_per_view"] = round(chain, 1)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v["us"], v["resident_waves_per_simd"], v["share"], v["hbm_GBs"]) for k, v in res["kernels"].items()}), "chain", res["chain_us_per_view"])
