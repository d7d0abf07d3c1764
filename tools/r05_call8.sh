#!/bin/bash
# round 5, GPU call 8 (developer script): convx K-split for small launches - tests, then small-batch bench vs the previous build policy
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_normfold.py tests/test_gpu_harness.py tests/test_gpu_configs.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/status.txt
for m in l32 l20; do for b in 1 2 4 8 16 32; do for opt in "" "--opt scan_segments=0"; do
  timeout 300 python bench.py --model $m --batch $b --steps 30 --warmup 10 --cpu-seqs 0 --host-seqs 0 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m b=$b [$opt]', round(d['value'],1), 'seq/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],4) for k,v in d.get('kernels',{}).items()})" >> $O/small_batch.txt
done; done; done
timeout 300 python bench.py --dtype f32 --opt f32_gemm_split=1 --batch 1 --steps 30 --warmup 10 --cpu-seqs 0 --host-seqs 0 > $O/bench_f32split_b1.json 2>/dev/null
timeout 900 python3 tools/fuzz_model.py 80 11 opts > $O/fuzz_opts.txt 2>&1; echo "fuzz rc=$?" >> $O/status.txt
echo "all done" >> $O/status.txt
