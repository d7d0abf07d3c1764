#!/usr/bin/env python3
"""Developer tool (no GPU needed): static instruction census of a kernel's hottest loop from hipcc's gfx950 assembly.

    python tools/isa_census.py plantcaduceus_amd/csrc/convx.hip 'convx_kernelItLb1ELi6E'      # mangled-name substring
    python tools/isa_census.py plantcaduceus_amd/csrc/scan.hip  'scan_kernelItLb1ELi2ELb1ELb1ELi64ELb1ELi0ELb1E'
    python tools/isa_census.py --check-res-waits [plantcaduceus_amd/csrc/gemm.hip]     # gemm.hip's hand-counted waits vs the emitted order

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


VMEM_RE = re.compile(r"^(buffer_|global_|flat_|scratch_)")


def _kernel_instrs(txt, name):
    body = txt[txt.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    out = []
    for ln in body:
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        out.append(s)
    return out


def _regs(tok):
    """'a[4:7]' -> ('a', {4,5,6,7}); 'v17' -> ('v', {17}); else None"""
    m = re.match(r"^([av])\[(\d+):(\d+)\]$", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"^([av])(\d+)$", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


def check_res_waits(src):
    """Post-build check of gemm.hip's hand-counted waits: in every gemm256q_kernel<.., EPI_RES / EPI_SCALE> the
    registers written by the un-waited inline-asm loads (64 x global_load_dwordx4 into accumulator quads; 8 x global_load_dword
    row factors) must not be READ by any instruction before an s_waitcnt whose vmcnt proves the load complete - counting, in
    program text order, the vector-memory operations issued after the load (loads of one type return in order on gfx950)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-S", "--cuda-device-only",
                        src, "-o", out], check=True, capture_output=True)
        txt = open(out).read()
    names = [m for m in re.findall(r"^(_Z\w+):", txt, flags=re.M) if "gemm256q_kernel" in m and ("Li2EEE" in m or "Li1EEE" in m)]
    if not names:
        sys.exit("no gemm256q_kernel<.., EPI_SCALE / EPI_RES> found")
    bad = 0
    for name in names:
        ins = _kernel_instrs(txt, name)
        pending = {}          # (file, reg) -> index (in VM-op order) of the load that writes it
        nvm = 0               # vector-memory operations issued so far (text order)
        loads = reads_ok = 0
        viol = []
        for s in ins:
            op = s.split()[0]
            ops = [t.strip() for t in s[len(op):].split(",")]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", s)
                if m:
                    done_upto = nvm - int(m.group(1))          # VM ops with index < done_upto have completed
                    for k in [k for k, idx in pending.items() if idx < done_upto]:
                        del pending[k]
                continue
            is_asm_load = op in ("global_load_dwordx4", "global_load_dword") and ops and _regs(ops[0]) and \
                (op == "global_load_dword" or ops[0].startswith("a["))
            # reads: every operand except the destination of a load / the accumulator destination of an MFMA counts as read too
            # (v_mfma D, A, B, C with C == D reads D)
            srcs = ops[1:] if (VMEM_RE.match(op) and "load" in op) else (ops if op.startswith("v_mfma") else ops[1:])
            for t in srcs:
                r = _regs(t.split(" ")[0])
                if r:
                    for n in r[1]:
                        if (r[0], n) in pending:
                            viol.append(s)
                        elif op.startswith("v_mfma") or op.startswith("v_mul") or op.startswith("v_pk_mul"):
                            reads_ok += 1
            if VMEM_RE.match(op):
                if is_asm_load:
                    r = _regs(ops[0])
                    for n in r[1]:
                        pending[(r[0], n)] = nvm
                    loads += 1
                nvm += 1
        kind = "EPI_RES" if "Li2EEE" in name else "EPI_SCALE"
        print(f"{name[:60]}... {kind}: {loads} un-waited asm loads, {len(viol)} reads of a register before a covering s_waitcnt")
        for v in viol[:5]:
            print("   VIOLATION:", v)
        bad += len(viol)
        if loads == 0:
            print("   no asm loads found (pattern changed?)")
            bad += 1
    print("check-res-waits:", "OK" if bad == 0 else f"{bad} problems")
    return bad


def main():
    if "--check-res-waits" in sys.argv:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        sys.exit(1 if check_res_waits(args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                        "plantcaduceus_amd", "csrc", "gemm.hip")) else 0)
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
