#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite .db) -> compact per-kernel summary text for profiles/.

    python profiles/summarize.py gpurun_out/<run>/prof/<name>_results.db profiles/<round>_<what>.txt [note]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)                 # drop the argument list
    name = name.replace("void ", "").replace("unsigned short", "bf16")
    return name[:110]


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        if note:
            f.write("# " + note + "\n")
        f.write("%-112s %8s %14s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            if pct < 0.001:
                continue
            f.write("%-112s %8d %14.1f %10.2f %7.3f\n" % (short(name), calls, tot / 1e3 if tot > 1e7 else tot, avg, pct))
    print(open(out).read())


if __name__ == "__main__":
    main()
