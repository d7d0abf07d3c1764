#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fulldepth.py > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/status.txt
timeout 900 python3 tools/fuzz_model.py 80 13 opts > $O/fuzz_opts.txt 2>&1; echo "fuzz rc=$?" >> $O/status.txt
echo "all done" >> $O/status.txt
