#!/bin/bash
# round 3, GPU call 6: GEMM epilogue experiments (staggered block phases, 64-byte-contiguous stores) + rest of the GPU suite
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03f"; mkdir -p "$O"; cd "$ROOT"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
for r in 1 2; do
  timeout 300 python3 bench.py $B 2>&1 | show "cur" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_GEMM_STAGGER=430 timeout 300 python3 bench.py $B 2>&1 | show "stagger430" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_GEMM_STAGGER=215 timeout 300 python3 bench.py $B 2>&1 | show "stagger215" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_GEMM_EPI_SWAP=1 timeout 300 python3 bench.py $B 2>&1 | show "episwap" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_GEMM_STAGGER=860 timeout 300 python3 bench.py $B 2>&1 | show "stagger860" | tee -a "$O/ab.txt"
done
PCAD_DEV=1 PCAD_GEMM_EPI_SWAP=1 PCAD_GEMM_STAGGER=430 timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear" 2>&1 | tail -3 | tee "$O/gemm_tests.txt"
timeout 1500 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fulldepth.py::test_full_depth_fp32 --deselect tests/test_gpu_fulldepth.py::test_full_depth_bf16_both_orders 2>&1 | tail -8 | tee "$O/gpu_tests.txt"
