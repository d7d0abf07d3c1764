#!/bin/bash
# round 5, GPU call 7 (developer script): small batches - scan segment policy with up to 16 segments of >= 32 steps (variants/libpcad_seg16.so) vs shipped
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/plantcaduceus_amd/variants/libpcad_seg16.so
for m in l32 l20; do for b in 1 2 4 8 16; do for lib in base seg16; do
  if [ $lib = base ]; then E=""; else E="PCAD_ALLOW_STALE=1 PCAD_LIB=$V"; fi
  env $E timeout 300 python bench.py --model $m --batch $b --steps 30 --warmup 10 --cpu-seqs 0 --host-seqs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m b=$b $lib', round(d['value'],1), 'seq/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],4) for k,v in d.get('kernels',{}).items()})" >> $O/small_batch.txt
done; done; done
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --durations=20 > $O/durations.log 2>&1
echo "all done" >> $O/status.txt
