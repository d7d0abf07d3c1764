#!/bin/bash
# round 6, GPU call 4: the waterfall-free wrap-around GEMM, SPLITY without contraction; strand-stride experiment for the 8 192-bp scan
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06d}; mkdir -p $O
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg"
timeout 900 python3 -m pytest tests/test_gpu_fulldepth.py -m gpu -q -x -s -k "split" 2>&1 | tail -8 > $O/tests_fulldepth_split.log
timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x -k "split or shortcut" 2>&1 | tail -5 > $O/tests_split.log
for c in 0 256; do
  timeout 400 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 4 --warmup 2 --chunk-seqs $c $B > $O/bench_f32_split_chunk$c.json 2>> $O/err.txt
done
for L in 8192 8200 8320 9216; do
  timeout 400 python3 bench.py --model pc2-medium --seqlen $L --batch 32 --steps 3 --warmup 1 $B > $O/bench_pc2m_L$L.json 2>> $O/err.txt
done
timeout 400 python3 bench.py --model pc2-medium --seqlen 8192 --batch 31 --steps 3 --warmup 1 $B > $O/bench_pc2m_L8192_b31.json 2>> $O/err.txt
timeout 400 python3 bench.py --model pc2-medium --seqlen 8192 --batch 16 --steps 3 --warmup 1 $B > $O/bench_pc2m_L8192_b16.json 2>> $O/err.txt
timeout 400 python3 bench.py --model pc2-medium --seqlen 512 --batch 512 --steps 3 --warmup 1 $B > $O/bench_pc2m_L512.json 2>> $O/err.txt
python3 - $O <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]["rows_per_launch"]
        k = {n: round(v["avg_ms"] * 1e6 / r, 2) for n, v in d.get("kernels", {}).items()}
        tok = d["value"] * d["config"]["seq_len"]
        print(f"{os.path.basename(f):34s} {d['value']:8.1f} seq/s {tok/1e3:8.1f} ktok/s {d['ms_per_step']:9.2f} ms rows/launch {r}  ns/row: {k}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
PY
tools/kpmc.sh scan_kernel r06d_pc2m_8192 --model pc2-medium --seqlen 8192 --batch 32 > /dev/null 2>&1
cat $O/tests_fulldepth_split.log $O/tests_split.log
