#!/bin/bash
# round 3, GPU call 4: 16-wave convx (swizzled / plain raw tile) correctness + A/B
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03d"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
PCAD_DEV=1 PCAD_CONVX16=1 timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_xproj" 2>&1 | tail -4 | tee "$O/cx16_test.txt"
PCAD_DEV=1 PCAD_CONVX16=1 PCAD_LIB="$V/libpcad_cx16noswz.so" timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_xproj" 2>&1 | tail -4 | tee -a "$O/cx16_test.txt"
for r in 1 2; do
  timeout 300 python3 bench.py $B 2>&1 | show "cur" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_CONVX16=1 timeout 300 python3 bench.py $B 2>&1 | show "cx16swz" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_CONVX16=1 PCAD_LIB="$V/libpcad_cx16noswz.so" timeout 300 python3 bench.py $B 2>&1 | show "cx16noswz" | tee -a "$O/ab.txt"
done
PCAD_DEV=1 PCAD_CONVX16=1 timeout 600 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4 | tee "$O/cx16_model_test.txt"
