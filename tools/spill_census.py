#!/usr/bin/env python3
"""developer tool: where do a kernel's scratch (spill) accesses sit?  Per basic block of every kernel whose mangled name contains
one of the given substrings: number of scratch_ instructions, v_exp_f32 (marks the scan's step loop) and MFMAs.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only x.hip -o /tmp/x.s ; tools/spill_census.py /tmp/x.s <substr> ..."""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pats = sys.argv[2:]
    name, blocks, cur = None, [], None

    def flush():
        if name and (not pats or any(p in name for p in pats)):
            tot = sum(b[1] for b in blocks)
            print(name[:110], "scratch ops:", tot)
            for b in blocks:
                if b[1] or b[2] > 8:
                    print("    %-12s scratch %3d  v_exp %3d  mfma %3d  lines %4d" % tuple(b))
    for ln in txt:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            flush()
            name, blocks, cur = m.group(1), [], ["entry", 0, 0, 0, 0]
            blocks.append(cur)
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", ln)
        if m and name:
            cur = [m.group(1), 0, 0, 0, 0]
            blocks.append(cur)
            continue
        if cur is None:
            continue
        if "scratch_" in ln:
            cur[1] += 1
        if "v_exp_f32" in ln:
            cur[2] += 1
        if "v_mfma" in ln:
            cur[3] += 1
        cur[4] += 1
    flush()


if __name__ == "__main__":
    main()
