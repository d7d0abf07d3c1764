#!/bin/bash
# round 6, GPU call 3: fp32 + f32_gemm_split at different chunk sizes on the wrap-around build (is the per-row time a function of the launch size?)
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06c}; mkdir -p $O
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg"
for c in 256 342 128 512; do
  timeout 400 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 3 --warmup 1 --chunk-seqs $c $B > $O/bench_f32_split_chunk$c.json 2>> $O/err.txt
done
timeout 400 python3 bench.py --dtype f32 --steps 2 --warmup 1 $B > $O/bench_f32_plain.json 2>> $O/err.txt
python3 tools/gemm_stride_probe.py 350208 > $O/stride_probe_350208.txt 2>&1
python3 - $O <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]["rows_per_launch"]
        k = {n: round(v["avg_ms"] * 1e6 / r, 2) for n, v in d.get("kernels", {}).items()}
        print(f"{os.path.basename(f):36s} {d['value']:8.1f} seq/s {d['ms_per_step']:9.2f} ms rows/launch {r}  ns/row: {k}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
PY
cat $O/stride_probe_350208.txt
