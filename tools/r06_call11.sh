#!/bin/bash
# round 6, GPU call 11: what the in-kernel dt_proj tile costs the scan (ablation builds), fp32 split and bf16, same box
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06l}; mkdir -p $O
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-10s %-16s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj','add_rmsnorm','gemm_out_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 2 --warmup 1 --profile-stride 1"
for n in cur notile hotboth notilehot cur; do
  if [ "$n" = cur ]; then L=""; else L="$V/libpcad_$n.so"; fi
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --dtype f32 --opt f32_gemm_split=1 $B 2>>$O/err.txt | show $n l32_f32_split | tee -a $O/ab.txt
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py $B 2>>$O/err.txt | show $n l32_bf16 | tee -a $O/ab.txt
done
