"""Timeline of one rocprofv3 --kernel-trace CSV: busy union vs wall per bench step, largest idle gaps and what precedes them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    if "namespace)::" in n:
        n = n.split("namespace)::")[1]
    return n.split("(")[0].split("<")[0][-48:]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))
if "--names" in sys.argv: print(sorted(set(e[2] for e in ev)))
# steps: split at preprocess_fwd_views launches
starts = [i for i, e in enumerate(ev) if "preprocess_fwd_views" in e[2] or "surfel_preprocess_fwd" in e[2]]
if len(starts) > 4 and "surfel" in ev[starts[0]][2]:
    starts = starts[::4]
if "--every" in sys.argv:   # per-view call pattern: a step = every K-th single-view K1
    starts = [i for i, e in enumerate(ev) if "preprocess_fwd" in e[2]][::int(sys.argv[sys.argv.index("--every") + 1])]
print("kernels", len(ev), "steps", len(starts))
for si in range(1, len(starts) - 1):
    seg = ev[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][0], ev[starts[si + 1]][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    last_name = seg[0][2]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    gaps.append((t1 - cur_e, last_name, "next step K1"))
    print(f"step {si}: wall {(t1 - t0) / 1e3:.0f} us, GPU busy {busy / 1e3:.0f} us ({100 * busy / (t1 - t0):.1f} %), idle {(t1 - t0 - busy) / 1e3:.0f} us")
    for g, a, b in sorted(gaps, reverse=True)[:6]:
        print(f"     gap {g / 1e3:7.1f} us after {a} before {b}")
if "--timeline" in sys.argv and len(starts) > 2:   # one step, kernel by kernel: start (us from the step's K1), duration, queue
    qcol = next((c for c in ("Queue_Id", "Stream_Id", "Correlation_Id") if c in rows[0]), None)
    full = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get(qcol, "")) for r in rows))
    i0 = max(1, len(starts) - 3)     # a steady-state step (the last complete one but one), not the first after warm-up
    a, b = full[starts[i0]][0], full[starts[i0 + 1]][0]
    print(f"timeline of step {i0}:")
    for s, e, n, q in full[starts[i0]:starts[i0 + 1]]:
        print(f"{(s - a) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  q{q}  {n}")
