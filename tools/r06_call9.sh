#!/bin/bash
# round 6, GPU call 9: pair walks - tests, then 8 192-bp / medium-batch bench lines vs scan_segments=0 on one box
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06i}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_normfold.py -m gpu -q -x 2>&1 | tail -25 > $O/tests_model.log
echo "tests_model rc=${PIPESTATUS[0]}" >> $O/status.txt
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 3 --warmup 1"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-9s %-22s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
for o in "" "--opt scan_segments=0"; do
  t=$(echo "$o" | tr -c 'a-z0-9=_' '_')
  timeout 300 python3 bench.py --model pc2-medium --seqlen 8192 --batch 32 $B $o 2>>$O/err.txt | show "pair$t" pc2m_8192_b32 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --model pc2-large --seqlen 8192 --batch 32 $B $o 2>>$O/err.txt | show "pair$t" pc2l_8192_b32 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --model pc2-medium --seqlen 8192 --batch 16 $B $o 2>>$O/err.txt | show "pair$t" pc2m_8192_b16 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --batch 32 $B $o 2>>$O/err.txt | show "pair$t" l32_b32 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --batch 16 $B $o 2>>$O/err.txt | show "pair$t" l32_b16 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --model l20 --batch 64 $B $o 2>>$O/err.txt | show "pair$t" l20_b64 | tee -a $O/ab.txt
  timeout 300 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --batch 32 $B $o 2>>$O/err.txt | show "pair$t" l32_f32split_b32 | tee -a $O/ab.txt
done
cat $O/tests_model.log | tail -12; cat $O/status.txt
