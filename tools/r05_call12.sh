#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=$PWD/gpurun_out/r05k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "workspace_limit or bind_time or poisoned" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/status.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --dtype f32 --opt f32_gemm_split=1 --steps 3 --warmup 1 --cpu-seqs 0 --host-seqs 0 > $O/bench_f32_split_under_rocprof.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_f32_split.csv; rm -rf $O/stats
echo "all done" >> $O/status.txt
