#!/bin/bash
# round 3, GPU call 1: microbenchmarks, softplus accuracy, scan variants A/B, scan PMC
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03a"; mkdir -p "$O"; cd "$ROOT"
V="$ROOT/plantcaduceus_amd/variants"
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_microbench tools/valu_microbench.hip 2>/dev/null && /tmp/valu_microbench > "$O/valu_microbench.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I plantcaduceus_amd/csrc -o /tmp/softplus_test tools/softplus_test.hip 2>/dev/null && /tmp/softplus_test > "$O/softplus.txt" 2>&1
cat "$O/softplus.txt"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k:round(v.get('avg_ms',0),4) for k,v in d.get('kernels',{}).items()})"; }
for r in 1 2; do for n in base new pp hot; do
  PCAD_LIB="$V/libpcad_$n.so" timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-seqs 0 2>/dev/null | show "$n" | tee -a "$O/ab.txt"
done; done
for n in new pp; do
  PCAD_LIB="$V/libpcad_$n.so" timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "scan or softplus" 2>&1 | tail -3 | tee -a "$O/tests_$n.txt"
done
PCAD_LIB="$V/libpcad_pp.so" timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fulldepth.py -x -q -m gpu 2>&1 | tail -5 | tee -a "$O/tests_pp_model.txt"
PCAD_LIB="$V/libpcad_base.so" tools/kpmc.sh scan_kernel base > /dev/null 2>&1
PCAD_LIB="$V/libpcad_pp.so" tools/kpmc.sh scan_kernel pp > /dev/null 2>&1
cat gpurun_out/kpmc_base.txt gpurun_out/kpmc_pp.txt | head -120
