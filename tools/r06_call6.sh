#!/bin/bash
# round 6, GPU call 6: scan step-ordering variants (LDS read vs scalar loads sharing lgkmcnt; two y chains), same box
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06f}; mkdir -p $O
V="$ROOT/plantcaduceus_amd/variants"
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 3 --warmup 1"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['rows_per_launch']
    print('%-8s %-16s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
for n in $2; do
  if [ "$n" = cur ]; then L=""; else L="$V/libpcad_$n.so"; fi
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --model pc2-medium --seqlen 8192 --batch 32 $B 2>>$O/err.txt | show $n pc2m_8192_b32 | tee -a $O/ab.txt
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --model pc2-medium --seqlen 512 --batch 512 $B 2>>$O/err.txt | show $n pc2m_512_b512 | tee -a $O/ab.txt
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --steps 5 --warmup 2 --cpu-seqs 0 --host-seqs 0 --no-parity-leg 2>>$O/err.txt | show $n l32_bf16 | tee -a $O/ab.txt
done
