#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (FETCH_SIZE/, WRITE_SIZE/, SQ/ sub-directories of one run) -> profiles/<round>_pmc_traffic.json

    python profiles/pmc_summary.py gpurun_out/<run> profiles/<round>_pmc_traffic.json
"""
import json
import re
import sys

import pandas as pd

NOTE_SRC = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / TCC_EA0_RDREQ_* (separate passes) -- python3 bench.py --steps 1 "
            "--warmup 1 --cpu-seqs 0 --no-profile --batch {batch}  (l32 bf16, {rows} token-rows per launch)")
NOTE_CORR = ("per MI355X_MICROARCH.md §HBM: FETCH_SIZE on gfx950 reports 1/2 of a wide coalesced read (checked: conv and add_rmsnorm raw "
             "values are 0.52x / 0.49x their algorithmic read bytes); WRITE_SIZE is 1:1 (scan = rows*E*2 exactly). traffic_bytes = "
             "2*FETCH_SIZE + WRITE_SIZE. The scan's 2-byte-per-lane loads are outside the calibrated width, so its read side is an "
             "upper estimate.")


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("unsigned short", "bf16")


def main():
    base, out = sys.argv[1].rstrip("/") + "/", sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 524288        # token-rows per launch of the profiled run (--batch 1024: 524288)
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        df = pd.read_csv(base + f"{c}/p_counter_collection.csv")
        df = df[df.Counter_Name == c]
        df["k"] = df.Kernel_Name.map(short)
        for k, v in df.groupby("k").Counter_Value.mean().items():
            if k.startswith("pcad::") and "pack" not in k:
                res.setdefault(k, {})[c + "_KB"] = round(float(v), 1)
    import os
    for sub in ("RDREQ", "DRAM"):                      # optional passes: read requests by size, DRAM-side request counts
        fn = base + f"{sub}/p_counter_collection.csv"
        if not os.path.exists(fn):
            continue
        df = pd.read_csv(fn)
        df["k"] = df.Kernel_Name.map(short)
        p = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
        for k, row in p.iterrows():
            if k in res:
                res[k].update({c: float(row[c]) for c in p.columns})
                if sub == "RDREQ":
                    n32, n64, n128, tot = (row.get(c, 0.0) for c in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum",
                                                                      "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"))
                    res[k]["read_bytes_by_size"] = float(32 * n32 + 64 * n64 + 128 * n128)
                    res[k]["read_bytes_if_rest_64B"] = float(32 * n32 + 128 * n128 + 64 * max(0.0, tot - n32 - n128))
    df = pd.read_csv(base + "SQ/p_counter_collection.csv")
    df["k"] = df.Kernel_Name.map(short)
    p = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    for k, row in p.iterrows():
        if k in res:
            res[k].update({c: float(row[c]) for c in p.columns})
            gui = row["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
            d = df[(df.k == k) & (df.Counter_Name == "GRBM_GUI_ACTIVE")]
            dur = float((d.End_Timestamp - d.Start_Timestamp).mean())            # ns, the profiled dispatches themselves
            res[k]["profiled_duration_us"] = round(dur / 1e3, 2)
            res[k]["effective_clock_GHz"] = round(gui / dur, 3)                  # shader clock during the (profiled) dispatch
            res[k]["valu_busy_frac"] = round(row["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui, 3)   # quad-cycles
            res[k]["mfma_busy_frac"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui, 3)
    cls = {"selective_scan": [k for k in res if "scan_kernel" in k],
           # 4-wave GEMM by fused epilogue (gemm.hip EPI_*): <T, T, 1> = in_proj with the row scale, <T, T, 2> = out_proj + residual
           # (norm_fold); <T, T, 0> / round-3 names without the parameter = both projections of the reference-order path
           "gemm_in_proj": [k for k in res if "gemm256q" in k and k.endswith(", 1>")],
           "gemm_out_proj_res": [k for k in res if "gemm256q" in k and k.endswith(", 2>")],
           "gemm_in_out_proj": [k for k in res if "gemm256" in k and not k.endswith(", 1>") and not k.endswith(", 2>")]
                               or [k for k in res if "gemm_nt_kernel" in k and k.endswith("false>")],
           "rstd_reduce": [k for k in res if "rstd_kernel" in k],
           "gemm_x_proj": [k for k in res if "gemm_nt_kernel" in k and k.endswith("true>")],
           "conv1d_bidir": [k for k in res if "conv_bidir" in k],
           "conv_xproj_fused": [k for k in res if "convx" in k],
           "add_rmsnorm": [k for k in res if "add_rmsnorm" in k and (k.endswith("false>") or k.endswith("false, false>"))]}
    o = {"source": NOTE_SRC.format(batch=rows // 512, rows=rows), "correction": NOTE_CORR, "rows_per_launch": rows, "kernels": res, "classes": {}}
    try:
        o["src_hash"] = open(base + "src_hash.txt").read().strip()     # bench.source_hash() of the profiled build
    except OSError:
        pass
    for c, ks in cls.items():
        ks = [k for k in ks if "FETCH_SIZE_KB" in res[k] and "WRITE_SIZE_KB" in res[k]]
        # a default bench.py run also executes its parity_config leg (the fp32 + f32_gemm_split model: <float, ...> / <bf16, float, ...>
        # instantiations): the class figures are the HEADLINE's, i.e. the bf16-storage instantiations only, whenever there are any
        head = [k for k in ks if "<bf16" in k and "<bf16, float" not in k]
        if head:
            ks = head
        if not ks:
            continue
        f = sum(res[k]["FETCH_SIZE_KB"] for k in ks) / len(ks)
        w = sum(res[k]["WRITE_SIZE_KB"] for k in ks) / len(ks)
        rb = [res[k]["read_bytes_by_size"] for k in ks if "read_bytes_by_size" in res[k]]
        read_b = sum(rb) / len(rb) if rb else 2 * f * 1024        # size-resolved request counts when collected, else 2x FETCH_SIZE
        o["classes"][c] = {"traffic_bytes_per_launch": round(read_b + w * 1024), "fetch_raw_bytes": round(f * 1024),
                           "read_bytes": round(read_b), "read_bytes_source": "TCC_EA0_RDREQ_{32B,64B,128B}" if rb else "2 x FETCH_SIZE",
                           "write_bytes": round(w * 1024),
                           "valu_busy_frac": round(sum(res[k].get("valu_busy_frac", 0) for k in ks) / len(ks), 3),
                           "mfma_busy_frac": round(sum(res[k].get("mfma_busy_frac", 0) for k in ks) / len(ks), 3),
                           "effective_clock_GHz": round(sum(res[k].get("effective_clock_GHz", 0) for k in ks) / len(ks), 3)}
    json.dump(o, open(out, "w"), indent=1)
    print(json.dumps(o["classes"], indent=1))


if __name__ == "__main__":
    main()
