import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
qcol = next((c for c in ("Queue_Id", "Stream_Id") if c in rows[0]), None)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qcol, "")) for r in rows)
k1 = [i for i, e in enumerate(ev) if "preprocess_fwd_views" in e[2]]
for si in range(1, len(k1) - 1):
    seg = ev[k1[si]:k1[si + 1]]
    t0 = seg[0][1]   # K1 end
    firsts = {}
    k6 = {}
    for s, e, n, q in seg:
        if "tile_count" in n and q not in firsts: firsts[q] = (s - t0) / 1e3
        if "render_fwd_kernel" in n and q not in k6: k6[q] = ((s - t0) / 1e3, (e - t0) / 1e3)
    k7 = [(s - t0) / 1e3 for s, e, n, q in seg if "render_bwd" in n]
    print("step", si, "K1 dur %.0f" % ((seg[0][1] - seg[0][0]) / 1e3), "tile_count start after K1 end by queue", {q: round(v) for q, v in firsts.items()},
          "K6 (start,end)", {q: (round(a), round(b)) for q, (a, b) in k6.items()}, "first K7 at", round(min(k7)) if k7 else None)
