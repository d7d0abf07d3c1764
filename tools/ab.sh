#!/bin/bash
# same-box A/B: tools/ab.sh "<env A>" "<env B>" [rounds]   (prints value + per-kernel ms of each bench run)
A="$1"; B="$2"; R="${3:-2}"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
for i in $(seq $R); do
  env $A timeout 300 python bench.py --steps 6 --warmup 2 --cpu-seqs 0 2>/dev/null | show "A[$A]"
  env $B timeout 300 python bench.py --steps 6 --warmup 2 --cpu-seqs 0 2>/dev/null | show "B[$B]"
done
