"""Instruction mix of the innermost v_exp_f32 loop of every render kernel in a device assembly file.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S render.hip -o /tmp/r.s
    python scripts/isa_loop_count.py /tmp/r.s [kernel-name-substring ...]

Prints, per kernel, the VALU / DPP / packed / move / select / SALU / LDS / VMEM instruction counts of the block of
assembly between the loop header in front of the first v_exp_f32 and the back edge behind the last one (the 4x unrolled
inner loop of K6 / K7), and the kernel's VGPR count.  A measurement aid for DESIGN.md section 3 (K6 / K7 are VALU-issue
bound: instructions per iteration is the number that matters)."""
import re
import sys
from collections import Counter


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    parts = re.split(r"\n(?=_ZN3gdr[^\n]*: +; @)", txt)
    for k in parts:
        name = k.split(":")[0]
        if not name.startswith("_ZN3gdr") or "v_exp_f32" not in k:
            continue
        if want and not any(w in name for w in want):
            continue
        lines = k.split("\n")
        idx = [i for i, l in enumerate(lines) if "v_exp_f32" in l]
        lo, hi = idx[0], idx[-1]
        while lo > 0 and "Loop Header" not in lines[lo] and "Parent Loop" not in lines[lo]:
            lo -= 1
        label = lines[lo].split(":")[0].strip()     # the back edge: the last branch to the loop header's label
        back = [i for i, l in enumerate(lines) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", l) and i > hi]
        hi = back[0] if back else hi
        c = Counter()
        for l in lines[lo:hi + 1]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if not m:
                continue
            op = m.group(1)
            if op.startswith("v_"):
                c["valu"] += 1
                if "dpp" in l or "quad_perm" in l or "row_" in l:
                    c["dpp"] += 1
                if op.startswith("v_pk"):
                    c["pk"] += 1
                if op.startswith("v_mov"):
                    c["mov"] += 1
                if op.startswith("v_cndmask"):
                    c["cnd"] += 1
                if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")):
                    c["trans"] += 1
            elif op.startswith("s_nop"):
                c["nop"] += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["ds"] += 1
            elif op.startswith(("global", "buffer", "flat")):
                c["vmem"] += 1
        vg = re.search(r"next_free_vgpr (\d+)", k)
        short = re.sub(r"^_ZN3gdr12_GLOBAL__N_1\d+", "", name)[:34]
        print(f"{short:36s} exps={len(idx)} vgpr={vg.group(1) if vg else '?'} " + " ".join(f"{a}={b}" for a, b in sorted(c.items())))


if __name__ == "__main__":
    main()
