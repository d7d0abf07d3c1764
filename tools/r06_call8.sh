#!/bin/bash
# round 6, GPU call 8: PRE = 96 scan at 3 waves per SIMD (PlantCAD2 Large) vs 4; fp32 split after dropping its L2 prefetch
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/${1:-r06h}; mkdir -p $O
V="$ROOT/plantcaduceus_amd/variants"
B="--cpu-seqs 0 --host-seqs 0 --no-parity-leg --steps 3 --warmup 1"
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-9s %-16s' % ('$1','$2'), round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k in ('selective_scan','conv_xproj_fused','gemm_in_proj','add_rmsnorm','gemm_out_proj')})
except Exception as e: print('$1','$2','failed',e)"; }
for n in cur occ96 occ96ch8 cur; do
  if [ "$n" = cur ]; then L=""; else L="$V/libpcad_$n.so"; fi
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --model pc2-large --seqlen 8192 --batch 32 $B 2>>$O/err.txt | show $n pc2l_8192_b32 | tee -a $O/ab.txt
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 300 python3 bench.py --model pc2-large --seqlen 512 --batch 512 $B 2>>$O/err.txt | show $n pc2l_512_b512 | tee -a $O/ab.txt
done
timeout 300 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 3 --warmup 1 --cpu-seqs 0 --host-seqs 0 --no-parity-leg 2>>$O/err.txt | show cur l32_f32_split | tee -a $O/ab.txt
timeout 600 python3 bench.py --steps 5 --warmup 2 > $O/bench_default_with_parity_leg.json 2>>$O/err.txt
python3 -c "
import json
d=json.load(open('$O/bench_default_with_parity_leg.json'))
print('default bench:', d['value'], 'parity_config:', json.dumps(d.get('parity_config'))[:600])
print('cpu_baseline:', json.dumps({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})[:700])"
