#!/bin/bash
# developer tool (GPU box): interleaved same-box A/B of engine options / builds.   tools/ab_opts.sh <tag> <rounds> "<spec>" ...
#   spec = "name|ENV=.. ENV=..|bench args"   e.g.  "fold|PCAD_LIB=/x.so|--opt norm_fold=1"
#   ("r3" as name: bench.py of a `git worktree add r3tree <round-3 commit>` checkout with its own built libpcad.so, i.e. a whole
#    earlier tree on the same box; builds of THIS tree with extra flags come from tools/build_variant.sh and go in through PCAD_LIB,
#    with PCAD_ALLOW_STALE=1 when the variant was built before the last `make` refreshed build_hash.h)
TAG="$1"; R="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/$TAG"; mkdir -p "$O"; cd "$ROOT"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 6 --warmup 2 --cpu-seqs 0 --host-seqs 0"
for r in $(seq $R); do
  for spec in "$@"; do
    IFS='|' read -r name envs bargs <<< "$spec"
    dir="$ROOT"; [ "$name" = r3 ] && dir="$ROOT/r3tree"
    ( cd "$dir" && env $envs timeout 300 python3 bench.py $B $bargs 2>/dev/null | show "$name" ) | tee -a "$O/ab.txt"
  done
done
