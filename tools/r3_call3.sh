#!/bin/bash
# round 3, GPU call 3: asm I/O (one wait per chunk) vs compiler I/O; 16-wave convx; failed + new tests; e2e; host timings
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03c"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
B="--steps 4 --warmup 1 --cpu-seqs 0 --host-seqs 0"
PCAD_DEV=1 PCAD_CONVX16=1 timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_xproj" 2>&1 | tail -3 | tee "$O/cx16_test.txt"
PCAD_DEV=1 PCAD_CONVX16=1 PCAD_LIB="$V/libpcad_cx16noswz.so" timeout 300 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_xproj" 2>&1 | tail -3 | tee -a "$O/cx16_test.txt"
for r in 1 2; do
  PCAD_LIB="$V/libpcad_noaio.so" timeout 300 python3 bench.py $B 2>&1 | show "noaio" | tee -a "$O/ab.txt"
  timeout 300 python3 bench.py $B 2>&1 | show "cur(aio2)" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_CONVX16=1 timeout 300 python3 bench.py $B 2>&1 | show "cur+cx16swz" | tee -a "$O/ab.txt"
  PCAD_DEV=1 PCAD_CONVX16=1 PCAD_LIB="$V/libpcad_cx16noswz.so" timeout 300 python3 bench.py $B 2>&1 | show "cur+cx16noswz" | tee -a "$O/ab.txt"
done
timeout 900 python3 -m pytest tests/test_ism.py tests/test_gpu_dist.py tests/test_xgb.py tests/test_vcf_fasta.py tests/test_sharding.py tests/test_known_answer.py -x -q -m gpu 2>&1 | tail -8 | tee "$O/tests_rest.txt"
timeout 900 python3 -m pytest tests/test_gpu_fulldepth.py -x -q -m gpu -k harsh -s 2>&1 | tail -8 | tee "$O/tests_harsh.txt"
timeout 600 python3 tools/e2e_5000.py > "$O/e2e_5000.json" 2> "$O/e2e_5000.err"; tail -c 1500 "$O/e2e_5000.json"; tail -3 "$O/e2e_5000.err"
timeout 600 python3 bench.py --steps 5 --warmup 2 --cpu-seqs 0 > "$O/bench_host.json" 2> "$O/bench_host.err"; python3 -c "
import json; d=json.loads(open('$O/bench_host.json').read().strip().splitlines()[-1]); print(d['value'], d.get('host'))"
