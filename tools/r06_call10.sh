#!/bin/bash
# round 6, GPU call 10: full GPU suite on the pair-walk build; pair threshold A/B at 3 072 waves (PlantCAD2 Large, 32 x 8 192 bp)
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06j}; mkdir -p $O
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 3 --warmup 1"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-12s %-18s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
for w in 2560 3584 4608; do
  PCAD_DEV=1 PCAD_PAIR_MAX_WAVES=$w timeout 300 python3 bench.py --model pc2-large --seqlen 8192 --batch 32 $B 2>>$O/err.txt | show "max$w" pc2l_8192_b32 | tee -a $O/ab.txt
  PCAD_DEV=1 PCAD_PAIR_MAX_WAVES=$w timeout 300 python3 bench.py --batch 48 $B 2>>$O/err.txt | show "max$w" l32_b48 | tee -a $O/ab.txt
  PCAD_DEV=1 PCAD_PAIR_MAX_WAVES=$w timeout 300 python3 bench.py --batch 64 $B 2>>$O/err.txt | show "max$w" l32_b64 | tee -a $O/ab.txt
done
timeout 3000 python3 -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 > $O/gpu_tests.log
tail -5 $O/gpu_tests.log
