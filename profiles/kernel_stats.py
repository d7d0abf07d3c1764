#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) -> compact text summary for profiles/.

    python profiles/kernel_stats.py gpurun_out/<run>/kernel_stats.csv profiles/<round>_kernel_stats.txt "<note>"
"""
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("unsigned short", "bf16")
    return n[:108]


def main():
    src, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else ""
    df = pd.read_csv(src)
    if "--headline-only" in sys.argv:
        # a default bench.py run also executes its parity_config leg (fp32 + f32_gemm_split model): keep the headline's kernels
        # (bf16 storage) and recompute the percentages over them
        keep = ~(df["Name"].str.contains("<float") | df["Name"].str.contains("<unsigned short, float") | df["Name"].str.contains("split_rows|pack_split"))
        df = df[keep].copy()
        df["Percentage"] = 100.0 * df["TotalDurationNs"] / df["TotalDurationNs"].sum()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        if note:
            f.write("# " + note + "\n")
        f.write("%-110s %8s %14s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for _, r in df.iterrows():
            if r["Percentage"] < 0.001:
                continue
            f.write("%-110s %8d %14.1f %10.2f %7.3f\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"] / 1e3,
                                                         r["AverageNs"] / 1e3, r["Percentage"]))
    print(open(out).read())


if __name__ == "__main__":
    main()
