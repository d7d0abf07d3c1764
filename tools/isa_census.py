#!/usr/bin/env python3
"""Developer tool (no GPU needed): static instruction census of a kernel's hottest loop from hipcc's gfx950 assembly.

    python tools/isa_census.py plantcaduceus_amd/csrc/convx.hip 'convx_kernelItLb1ELi6E'      # mangled-name substring
    python tools/isa_census.py plantcaduceus_amd/csrc/scan.hip  'scan_kernelItLb1ELi2ELb1ELb1ELi64ELb1ELi0ELb1E'

Finds every loop (a label that a LATER branch jumps back to), takes the one with the most instructions that contains no inner
loop of its own unless --outer is given, and counts its instructions by issue class: packed / plain / transcendental VALU, MFMA,
LDS, VMEM (incl. LDS-DMA), SMEM, SALU, waits / barriers.  The counts are per loop iteration (what one iteration is - a K-tile
pair, eight scan steps - is printed from the source's own structure by the caller's note)."""
import collections
import os
import re
import subprocess
import sys
import tempfile


def classify(op: str) -> str:
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f16", "v_rcp_f16"):
        return "VALU transcendental"
    if op.startswith("v_pk_"):
        return "VALU packed"
    if op.startswith("v_accvgpr"):
        return "VALU accvgpr move"
    if op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "VALU cross-lane"
    if op.startswith("v_cvt"):
        return "VALU convert"
    if op.startswith("v_"):
        return "VALU plain"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM (" + ("LDS-DMA" if False else "load/store") + ")"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op in ("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio"):
        return "wait / barrier / nop"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    src, key = sys.argv[1], sys.argv[2]
    outer = "--outer" in sys.argv
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-S", "--cuda-device-only",
                        src, "-o", out] + [a for a in sys.argv[3:] if a.startswith("-D")], check=True, capture_output=True)
        txt = open(out).read()
    names = [m for m in re.findall(r"^(_Z\w+):", txt, flags=re.M) if key in m]
    if not names:
        sys.exit(f"no kernel matching {key!r}")
    name = names[0]
    body = txt[txt.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    labels = {}
    instr = []                       # (index, op, line)
    for ln in body:
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(instr)
            continue
        if s.endswith(":"):
            continue
        op = s.split()[0]
        instr.append((op, s))
    loops = []
    for i, (op, s) in enumerate(instr):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    if not loops:
        sys.exit("no loop found")
    def has_inner(lp):
        return any(a > lp[0] and b < lp[1] for a, b in loops if (a, b) != lp)
    cand = [lp for lp in loops if outer or not has_inner(lp)] or loops
    a, b = max(cand, key=lambda lp: lp[1] - lp[0])
    cnt = collections.Counter(classify(op) for op, _ in instr[a:b + 1])
    lds_dma = sum(1 for op, s in instr[a:b + 1] if (op.startswith("buffer_load") or op.startswith("global_load")) and " lds" in s)
    print(f"kernel {name}")
    print(f"hottest {'outer ' if outer else 'innermost '}loop: {b - a + 1} instructions per iteration ({len(loops)} loops in the kernel)")
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v:5d}")
    if lds_dma:
        print(f"  (of the VMEM instructions, LDS-DMA: {lds_dma})")
    ops = collections.Counter(op for op, _ in instr[a:b + 1])
    print("  most frequent opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(14)))


if __name__ == "__main__":
    main()
