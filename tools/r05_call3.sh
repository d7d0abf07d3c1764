#!/bin/bash
# round 5, GPU call 3 (developer script): 4-wave GEMM with the LDS-DMA slot staggered per wave (variants/libpcad_stagger.so) vs the shipped build
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/plantcaduceus_amd/variants/libpcad_stagger.so
for r in 1 2 3; do
  echo "== base r$r" >> $O/gemm_time.txt;    python tools/gemm_time.py 524288 --no-vendor >> $O/gemm_time.txt 2>&1
  echo "== stagger r$r" >> $O/gemm_time.txt; PCAD_ALLOW_STALE=1 PCAD_LIB=$V python tools/gemm_time.py 524288 --no-vendor >> $O/gemm_time.txt 2>&1
done
PCAD_ALLOW_STALE=1 PCAD_LIB=$V timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_normfold.py -m gpu -q -x -k "linear or norm_fold or fold" > $O/tests_stagger.log 2>&1; echo "tests_stagger rc=$?" >> $O/status.txt
bash tools/ab_opts.sh r05c 3 "base||" "stagger|PCAD_ALLOW_STALE=1 PCAD_LIB=$V|" > $O/ab_stdout.txt 2>&1
echo "all done" >> $O/status.txt
