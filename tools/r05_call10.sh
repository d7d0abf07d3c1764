#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in l32 l20; do for b in 8 12 16 24 32; do for t in 64 128 256; do
  PCAD_DEV=1 PCAD_CX_SPLIT_TILES=$t timeout 300 python bench.py --model $m --batch $b --steps 30 --warmup 10 --cpu-seqs 0 --host-seqs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m b=$b max_tiles=$t', round(d['value'],1), 'seq/s', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_ms'],4) for k,v in d.get('kernels',{}).items() if k in ('conv_xproj_fused','selective_scan')})" >> $O/ksplit_tiles.txt
done; done; done
echo "all done" >> $O/status.txt
