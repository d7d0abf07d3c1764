#!/bin/bash
# round 6, GPU call 2 (developer script, run through gpurun): first run of the wrap-around K cursor ([hi | lo] split operands), the
# XCD-affine scan block order and the scan's L2 prefetch: the tests that cover them, then A/B-able bench lines on this box.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06b}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_normfold.py -m gpu -q -x 2>&1 | tail -15 > $O/tests_ops_model.log
echo "tests_ops_model rc=${PIPESTATUS[0]}" >> $O/status.txt
timeout 1500 python3 -m pytest tests/test_gpu_fulldepth.py -m gpu -q -x -s -k "split or fp32" 2>&1 | tail -15 > $O/tests_fulldepth_split.log
echo "tests_fulldepth rc=${PIPESTATUS[0]}" >> $O/status.txt
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg"
timeout 300 python3 bench.py --steps 8 --warmup 3 $B > $O/bench_l32_bf16.json 2>> $O/err.txt
timeout 400 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 4 --warmup 2 $B > $O/bench_l32_f32_split.json 2>> $O/err.txt
for spec in "pc2-medium 8192 32" "pc2-medium 512 512" "pc2-large 8192 32" "pc2-large 512 512"; do
  set -- $spec
  timeout 400 python3 bench.py --model $1 --seqlen $2 --batch $3 --steps 4 --warmup 2 $B > $O/bench_$1_$2.json 2>> $O/err.txt
done
timeout 300 python3 bench.py --steps 8 --warmup 3 $B > $O/bench_l32_bf16_r2.json 2>> $O/err.txt
timeout 400 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 4 --warmup 2 $B > $O/bench_l32_f32_split_r2.json 2>> $O/err.txt
python3 - $O <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = {n: v["avg_ms"] for n, v in d.get("kernels", {}).items()}
        print(f"{os.path.basename(f):40s} {d['value']:9.1f} seq/s {d['ms_per_step']:9.2f} ms  {k}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
PY
cat $O/status.txt
