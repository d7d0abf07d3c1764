#!/bin/bash
# round 3, GPU call 10: conv+x_proj window reads (one 16-byte read per row vs two 8-byte), power probe on the right card
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03j"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" 2>&1 | tail -3 | tee "$O/conv_test.txt"
for r in 1 2 3; do
  PCAD_LIB="$V/libpcad_cxb64.so" timeout 300 python3 bench.py $B 2>&1 | show "convx-b64" | tee -a "$O/ab.txt"
  timeout 300 python3 bench.py $B 2>&1 | show "convx-b128(cur)" | tee -a "$O/ab.txt"
done
timeout 300 python3 tools/power_probe.py > "$O/power_probe.txt" 2> "$O/power_probe.err"; cat "$O/power_probe.txt"; tail -3 "$O/power_probe.err"
