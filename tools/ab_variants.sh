#!/bin/bash
# developer tool (GPU box): interleaved same-box A/B of named builds (tools/build_variant.sh) against the in-tree libpcad.so
#   gpurun -- 'tools/ab_variants.sh <tag> <rounds> name1 name2 ...'      ("cur" = the in-tree build)
TAG="$1"; R="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/$TAG"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
for r in $(seq $R); do
  for n in "$@"; do
    if [ "$n" = cur ]; then timeout 300 python3 bench.py $B 2>&1 | show "cur" | tee -a "$O/ab.txt"
    else PCAD_LIB="$V/libpcad_$n.so" timeout 300 python3 bench.py $B 2>&1 | show "$n" | tee -a "$O/ab.txt"; fi
  done
done
