#!/usr/bin/env python
"""K7 / K6 phase budget from scripts/gpu_k7_budget.sh's runs (round-5 verdict next #1a): every number is MEASURED — launch
time from HIP events on the launch stream (bench.py's serial per-kernel pass), instruction counts from rocprofv3 --pmc —
on measurement builds of render.hip with one phase compiled out (GDR_K7_STUB).  A phase's share = product build - stub build.

    python scripts/k7_budget.py gpurun_out/k7_budget out.json [static_isa.json]
"""
import glob
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
static = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else None
MEANING = {
    "release": "product build",
    "k7stub1": "K7 record atomics never executed",
    "k7stub2": "K7 12-value DPP reduce-scatter -> 11 plain adds",
    "k7stub4": "K7 gradient terms behind dL/dalpha skipped",
    "k7stub8": "K7 slice cull (block_masks + row_lists_append) executed twice",
    "k7stub16": "K7 'behind' recurrences + dL/dalpha dot product skipped",
    "k7stub32": "K6 slice cull executed twice",
    "k7stub3": "K7 no atomics, no reduce-scatter",
    "k7stub7": "K7 no atomics, no reduce-scatter, no gradient terms",
    "k7stub23": "K7 walk skeleton: staging + cull + alpha / hit / T only",
}
res = {"note": "us = HIP events on the launch stream, kernel alone (bench.py serial pass), row kernel pinned (GDR_K7_PAIRS=0); "
               "valu / lds / wave_qcycles = rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAVE_CYCLES per launch of the "
               "kernel (all views of the node in one K7 launch; K6 per view).  Stub builds give WRONG gradients by construction.",
       "variants": MEANING, "workloads": {}}
for f in sorted(glob.glob(os.path.join(src, "time_*_*.json"))):
    m = re.match(r"time_(.+)_(c\d\w*)\.json", os.path.basename(f))
    try:
        d = json.load(open(f))
    except Exception:
        continue
    v, wl = m.group(1), m.group(2)
    k = d.get("kernels") or {}
    row = dict(step_ms=d.get("ms_per_step"), views_per_s=d.get("value"),
               k7_us=(k.get("render_bwd") or {}).get("avg_us_serial") or (k.get("render_bwd") or {}).get("avg_us"),
               k7_us_in_step=(k.get("render_bwd") or {}).get("avg_us"),
               k6_us=(k.get("render_fwd") or {}).get("avg_us_serial") or (k.get("render_fwd") or {}).get("avg_us"))
    p = os.path.join(src, f"pmc_{v}_{wl}.json")
    if os.path.exists(p):
        c = json.load(open(p))
        for kn, short in (("render_bwd_kernel", "k7"), ("render_fwd_kernel", "k6")):
            if kn in c:
                row[short + "_valu"] = int(c[kn].get("SQ_INSTS_VALU", 0))
                row[short + "_salu"] = int(c[kn].get("SQ_INSTS_SALU", 0))
                row[short + "_lds"] = int(c[kn].get("SQ_INSTS_LDS", 0))
                row[short + "_wave_qcycles"] = int(c[kn].get("SQ_WAVE_CYCLES", 0))
                row[short + "_waves"] = int(c[kn].get("SQ_WAVES", 0))
                row[short + "_active_valu_qcycles"] = int(c[kn].get("SQ_ACTIVE_INST_VALU", 0))
    res["workloads"].setdefault(wl, {})[v] = row
for wl, t in res["workloads"].items():
    base = t.get("release")
    if not base:
        continue
    sh = {}

    def diff(v, key, sign=1.0):
        if v in t and t[v].get(key) is not None and base.get(key) is not None:
            return round(sign * (base[key] - t[v][key]), 1)
        return None

    for v, label in (("k7stub1", "atomics"), ("k7stub2", "reduce_scatter_minus_11_adds"), ("k7stub4", "gradient_terms"),
                     ("k7stub16", "behind_state"), ("k7stub23", "everything_but_skeleton")):
        sh[label] = dict(k7_us=diff(v, "k7_us"), k7_valu=diff(v, "k7_valu"))
    sh["k7_cull"] = dict(k7_us=diff("k7stub8", "k7_us", -1.0), k7_valu=diff("k7stub8", "k7_valu", -1.0))
    sh["k6_cull"] = dict(k6_us=diff("k7stub32", "k6_us", -1.0), k6_valu=diff("k7stub32", "k6_valu", -1.0))
    t["_phase_share_vs_release"] = sh
if static:
    res["static_isa_inner_loop_per_4_entries"] = static
json.dump(res, open(out, "w"), indent=1)
for wl, t in res["workloads"].items():
    print("==", wl)
    for v, r in t.items():
        print("  %-26s %s" % (v, json.dumps(r)))
