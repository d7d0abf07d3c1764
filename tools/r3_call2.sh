#!/bin/bash
# round 3, GPU call 2: asm row I/O A/B, last-layer shortcut A/B, full GPU suite
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03b"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
for r in 1 2; do
  PCAD_LIB="$V/libpcad_noaio.so" timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-seqs 0 2>&1 | show "noaio" | tee -a "$O/ab.txt"
  timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-seqs 0 2>&1 | show "cur" | tee -a "$O/ab.txt"
  timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-seqs 0 --opt last_layer_shortcut=0 2>&1 | show "cur-noshortcut" | tee -a "$O/ab.txt"
done
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee "$O/gpu_tests.txt"
