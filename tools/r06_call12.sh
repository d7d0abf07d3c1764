#!/bin/bash
# round 6, GPU call 12: split-bf16 GEMMs with the three products of a Ko-tile interleaved (L2 reuse) vs three passes over K, same box
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06m}; mkdir -p $O
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-6s %-16s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj','add_rmsnorm','gemm_out_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 3 --warmup 1"
for n in cur ilv cur ilv; do
  if [ "$n" = cur ]; then L=""; else L="$V/libpcad_$n.so"; fi
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --dtype f32 --opt f32_gemm_split=1 $B 2>>$O/err.txt | show $n l32_f32_split | tee -a $O/ab.txt
done
PCAD_ALLOW_STALE=1 PCAD_LIB="$V/libpcad_ilv.so" timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fulldepth.py -m gpu -q -x -k "split" 2>&1 | tail -4 | tee -a $O/ab.txt
