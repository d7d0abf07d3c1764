#!/bin/bash
# round 5, GPU call 19 (developer script): a 2 048-window census (statistics for the noise model; not a test fixture)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05v; mkdir -p $O
export OMP_WAIT_POLICY=passive HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from oracle import c_oracle; c_oracle.build()" > $O/build.log 2>&1
OMP_NUM_THREADS=${CENSUS_THREADS:-224} timeout 3300 python oracle/gen_census_golden.py --model l32 --n 2048 --n-eng 512 --n-plainc 64 --chunk 128 --out $O/census_l32_2048.npz > $O/census.log 2>&1
timeout 600 python tools/argmax_census.py --model l32 --fixture $O/census_l32_2048.npz --batch 1024 > $O/argmax_census_l32_2048.txt 2>&1
echo "all done" >> $O/status.txt
