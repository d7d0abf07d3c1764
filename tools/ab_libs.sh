#!/bin/bash
# developer tool (GPU box): same-box A/B of variant libraries (plantcaduceus_amd/variants/libpcad_<name>.so, PCAD_ALLOW_STALE=1) against the
# in-tree build on one bench.py configuration:   tools/ab_libs.sh <tag> "<names, 'cur' = in-tree>" <bench.py args...>
TAG=$1; NAMES=$2; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
V="$ROOT/plantcaduceus_amd/variants"
for n in $NAMES; do
  if [ "$n" = cur ]; then L=""; else L="$V/libpcad_$n.so"; fi
  PCAD_ALLOW_STALE=1 PCAD_LIB="$L" timeout 400 python3 bench.py --cpu-seqs 0 --host-seqs 0 --no-parity-leg "$@" 2>>$O/err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-10s' % '$n', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items() if k!='final_head'})
except Exception as e: print('$n','failed',e)" | tee -a $O/ab.txt
done
