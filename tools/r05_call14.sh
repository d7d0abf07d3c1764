#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/final_round.sh r05x
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05x/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r05x/status.txt
timeout 600 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --cpu-seqs 0 > gpurun_out/r05x/bench_l32_f32_split.json 2>> gpurun_out/r05x/err.txt
echo "all done" >> gpurun_out/r05x/status.txt
