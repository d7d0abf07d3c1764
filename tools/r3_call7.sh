#!/bin/bash
# round 3, GPU call 7: dt_proj weight tile re-read per block vs resident; full GPU suite
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03g"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
for r in 1 2 3; do
  PCAD_LIB="$V/libpcad_wres.so" timeout 300 python3 bench.py $B 2>&1 | show "wdt-resident" | tee -a "$O/ab.txt"
  timeout 300 python3 bench.py $B 2>&1 | show "wdt-reread(cur)" | tee -a "$O/ab.txt"
done
timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | tail -12 | tee "$O/gpu_tests.txt"
