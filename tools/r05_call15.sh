#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05w; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python3 tools/fuzz_model.py 150 21 opts > $O/fuzz_opts.txt 2>&1; echo "fuzz_opts rc=$?" >> $O/status.txt
timeout 900 python3 tools/fuzz_model.py 100 22 > $O/fuzz_plain.txt 2>&1; echo "fuzz_plain rc=$?" >> $O/status.txt
timeout 900 python3 tools/fuzz_model.py 60 23 fold > $O/fuzz_fold.txt 2>&1; echo "fuzz_fold rc=$?" >> $O/status.txt
timeout 600 python3 tools/soak.py 100 poison > $O/soak_poison.txt 2>&1; echo "soak rc=$?" >> $O/status.txt
python3 -c "import sys; sys.path.insert(0,'.'); import bench; print(bench.source_hash())" > $O/src_hash.txt
echo "all done" >> $O/status.txt
