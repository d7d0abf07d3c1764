#!/usr/bin/env python3
"""developer tool: turn what `tools/final_round.sh <tag>` left under gpurun_out/<tag> (+ gpurun_out/<tag>s, kpmc_<tag>*.txt) into the
committed files profiles/<tag>_* (bench lines, rocprofv3 kernel tables, PMC traffic, kpmc, test / fuzz / soak logs, the
8 192-bp vs 512-bp per-token table).      python tools/collect_round.py r06z"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        return True
    print("missing", src)
    return False


def py(*args):
    r = subprocess.run([sys.executable] + list(args), cwd=ROOT, capture_output=True, text=True)
    if r.returncode:
        print("FAILED", args, r.stderr[-800:])


cp(f"{G}/bench.json", f"{tag}_bench_l32_bf16.json")
cp(f"{G}/bench_driver_cmd.json", f"{tag}_bench_driver_cmd.json")
cp(f"{G}/bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json")
cp(f"{G}/gpu_tests.log", f"{tag}_gpu_tests.log")
cp(f"{G}/e2e_5000.json", f"{tag}_e2e_5000.json")
for m in ("l20", "l24", "l28"):
    cp(f"{G}/bench_{m}.json", f"{tag}_bench_{m}.json")
for sfx, name in (("", "scan"), ("_convx", "convx"), ("_gemm", "gemm")):
    cp(os.path.join(ROOT, "gpurun_out", f"kpmc_{tag}{sfx}.txt"), f"{tag}_kpmc_{name}.txt")
src_hash = open(f"{G}/src_hash.txt").read().strip() if os.path.exists(f"{G}/src_hash.txt") else "?"
py("profiles/kernel_stats.py", f"{G}/kernel_stats.csv", f"profiles/{tag}_kernel_stats.txt",
   f"final tree of the round (src_hash {src_hash}): rocprofv3 --kernel-trace --stats of bench.py --steps 3 --warmup 1 (1024 windows per step as 2 chunks of 512); headline kernels only", "--headline-only")
py("profiles/pmc_summary.py", G, f"profiles/{tag}_pmc_traffic.json")
if os.path.exists(f"{G}/kernel_stats_f32_split.csv"):
    py("profiles/kernel_stats.py", f"{G}/kernel_stats_f32_split.csv", f"profiles/{tag}_kernel_stats_f32_split.txt",
       "fp32 model with f32_gemm_split (the parity configuration): bench.py --dtype f32 --opt f32_gemm_split=1 --steps 3 --warmup 1 (1024 windows per step as 4 chunks of 256)")
    cp(f"{G}/bench_f32_split_under_rocprof.json", f"{tag}_bench_f32_split_under_rocprof.json")

# ---- item 2: 8 192 bp vs 512 bp per token, PlantCAD2 Medium / Large -------------------------------------------------------
import pandas as pd


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("unsigned short", "bf16").replace("pcad::", "")
    return n[:74]


out = [f"# {tag}: bench.py --model pc2-{{medium,large}} --seqlen 8192 --batch 32 next to --seqlen 512 --batch 512 (524 288 token-rows per launch both) under",
       "# rocprofv3 --kernel-trace --stats on the final kernels; ns per token = total kernel time / (windows x bp x 4 steps); round-5 kernels: r06_kernel_stats_pc2_8192_before.txt"]
for m in ("pc2-medium", "pc2-large"):
    rows, ok = {}, True
    for L, b in ((8192, 32), (512, 512)):
        f = f"{G}/kernel_stats_{m}_{L}.csv"
        if not os.path.exists(f):
            ok = False
            continue
        py("profiles/kernel_stats.py", f, f"profiles/{tag}_kernel_stats_{m.replace('-', '_')}_{L}.txt", f"bench.py --model {m} --seqlen {L} --batch {b} --steps 3 --warmup 1")
        df = pd.read_csv(f)
        tokens = b * L * 4
        for _, r in df.iterrows():
            if r["Percentage"] < 0.3:
                continue
            rows.setdefault(short(r["Name"]), {})[L] = (int(r["Calls"]), r["AverageNs"] / 1e3, r["TotalDurationNs"] / 1e3 / tokens * 1e3)
        rows.setdefault("TOTAL (all kernels)", {})[L] = (0, 0.0, df.TotalDurationNs.sum() / 1e3 / tokens * 1e3)
    if not ok:
        continue
    out.append(f"== {m}: kernel | calls  avg_us  ns/token @ 8 192 bp | calls  avg_us  ns/token @ 512 bp | ratio")
    for k, v in rows.items():
        a, c = v.get(8192, (0, 0, 0)), v.get(512, (0, 0, 0))
        out.append(f"{k:76s} {a[0]:5d} {a[1]:9.1f} {a[2]:8.2f} | {c[0]:5d} {c[1]:9.1f} {c[2]:8.2f} | {(a[2] / c[2]) if c[2] and a[2] else 0:5.2f}")
    for L in (8192, 512):
        j = f"{G}/bench_{m}_{L}_under_rocprof.json"
        if os.path.exists(j):
            d = json.loads(open(j).read().strip().splitlines()[-1])
            out.append(f"   bench line under the profiler, {L} bp: {d['value']:.1f} windows/s = {d['value'] * L / 1e3:.1f} k tokens/s, {d['ms_per_step']:.1f} ms/step")
open(os.path.join(P, f"{tag}_kernel_stats_pc2_summary.txt"), "w").write("\n".join(out) + "\n")

# ---- secondary lines ----------------------------------------------------------------------------------------------------
S = os.path.join(ROOT, "gpurun_out", tag + "s")
lines = [f"# secondary bench lines on the final tree of the round (one box; tools/refresh_secondary.sh {tag}s): PlantCAD2 geometries at 512 / 8 192 bp, fp32, small batches, the other workloads",
         "# tokens/s = windows/s x window length"]
for f in sorted(glob.glob(os.path.join(S, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        L = d["config"]["seq_len"]
        lines.append(f"{os.path.basename(f):36s} {d['value']:10.1f} {d['unit']:12s} {d['ms_per_step']:9.2f} ms/step  {d['dtype']}  {d['value'] * L / 1e3:9.1f} k tokens/s")
        if any(k in f for k in ("pc2-", "l32_f32")):
            shutil.copyfile(f, os.path.join(P, f"{tag}s_" + os.path.basename(f)))
    except Exception as ex:
        lines.append(f"{os.path.basename(f)} unreadable: {ex}")
open(os.path.join(P, f"{tag}s_summary.txt"), "w").write("\n".join(lines) + "\n")

# ---- fuzz / soak / smoke -------------------------------------------------------------------------------------------------
fz = [f"# {tag}: randomized model-level parity, race soak and smoke() on the final tree (src_hash {src_hash})"]
for name, what in (("fuzz_opts.txt", "tools/fuzz_model.py 120 606 opts (random geometry x random engine options)"), ("fuzz_plain.txt", "tools/fuzz_model.py 60 607"),
                   ("fuzz_fold.txt", "tools/fuzz_model.py 40 608 fold (norm-folded geometries)"), ("soak.txt", "tools/soak.py 100 poison"), ("smoke.txt", "__graft_entry__.smoke()")):
    f = f"{G}/{name}"
    if os.path.exists(f):
        body = [ln for ln in open(f).read().strip().splitlines() if ln.strip()]
        fails = [ln for ln in body if "FAIL" in ln]
        fz.append(f"== {what}")
        fz += ["   " + ln for ln in fails[:10]]
        fz.append("   " + (body[-1] if body else "(empty)"))
open(os.path.join(P, f"{tag}_fuzz_soak.txt"), "w").write("\n".join(fz) + "\n")
print(open(os.path.join(P, f"{tag}_kernel_stats_pc2_summary.txt")).read())
print(open(os.path.join(P, f"{tag}s_summary.txt")).read())
print(open(os.path.join(P, f"{tag}_fuzz_soak.txt")).read())
