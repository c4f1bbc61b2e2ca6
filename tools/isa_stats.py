#!/usr/bin/env python3
"""Instruction mix per basic block of a gfx950 assembly listing (hipcc -save-temps .s file).

    python tools/isa_stats.py file.s [kernel-name-substring]

Prints, for every kernel (or the ones whose mangled name contains the substring), the resource footer
(VGPRs, SGPRs, scratch, LDS) and for each basic block the count of VALU / SALU / LDS / VMEM / waitcnt /
barrier instructions -- the offline half of the profiling loop (no GPU needed)."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    kern = None
    blocks = []
    cur = None
    foot = {}
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", s)
        if m and not s.startswith(".") and kern is None and "@function" not in s:
            pass
        if s.startswith(".type") and "@function" in s:
            kern = s.split()[1].rstrip(",")
            blocks = []
            cur = None
            continue
        if kern is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith(kern + ":"):
            cur = {"name": s.split(":")[0], "n": {}}
            blocks.append(cur)
            continue
        if s.startswith(".amdhsa_next_free_vgpr") or s.startswith(".amdhsa_next_free_sgpr") \
                or s.startswith(".amdhsa_group_segment_fixed_size") or s.startswith(".amdhsa_private_segment_fixed_size") \
                or s.startswith(".amdhsa_accum_offset"):
            k, v = s.split()[:2]
            foot[k] = v
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".size") and kern in s:
            pass
        if s.startswith(".Lfunc_end"):
            if want in kern:
                print(f"== {kern}")
                tot = {}
                for b in blocks:
                    n = b["n"]
                    if sum(n.values()) == 0:
                        continue
                    for k, v in n.items():
                        tot[k] = tot.get(k, 0) + v
                    if sum(n.values()) >= 40:
                        print(f"  {b['name']:<14} " + " ".join(f"{k}={v}" for k, v in sorted(n.items())))
                print("  TOTAL          " + " ".join(f"{k}={v}" for k, v in sorted(tot.items())))
            kern_done = kern
            kern = None
            continue
        if cur is None or not s or s.startswith((".", ";")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur["n"][c] = cur["n"].get(c, 0) + 1
    # footers appear after the function bodies; print them all
    for line in open(path):
        s = line.strip()
        if s.startswith(".amdhsa_kernel "):
            name = s.split()[1]
            show = want in name
        if s.startswith((".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_group_segment_fixed_size",
                         ".amdhsa_private_segment_fixed_size")) and show:
            print(f"  [{name[:48]}] {s}")


if __name__ == "__main__":
    main()
