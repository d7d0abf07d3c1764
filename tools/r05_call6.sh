#!/bin/bash
# round 5, GPU call 6 (developer script): final_round (full GPU suite, profile round, PMC, other sizes, e2e) + option fuzz + soak
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:-r05z}
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/final_round.sh $TAG
O=gpurun_out/$TAG
timeout 2400 python3 tools/fuzz_model.py 120 5 opts > $O/fuzz_opts.txt 2>&1; echo "fuzz_opts rc=$?" >> $O/status.txt
timeout 900 python3 tools/fuzz_model.py 60 7 > $O/fuzz_plain.txt 2>&1; echo "fuzz_plain rc=$?" >> $O/status.txt
timeout 900 python3 tools/fuzz_model.py 40 9 fold > $O/fuzz_fold.txt 2>&1; echo "fuzz_fold rc=$?" >> $O/status.txt
timeout 900 python3 tools/soak.py > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/status.txt
timeout 600 python3 bench.py --dtype f32 --opt f32_gemm_split=1 > $O/bench_l32_f32_split.json 2>> $O/err.txt
timeout 600 python3 bench.py --opt reference_order=1 --cpu-seqs 0 > $O/bench_l32_reference_order1.json 2>> $O/err.txt
timeout 600 python3 bench.py --opt reference_order=2 --cpu-seqs 0 > $O/bench_l32_reference_order2.json 2>> $O/err.txt
timeout 600 python tools/argmax_census.py --model l32 --fixture tests/golden/census_l32.npz > $O/argmax_census_l32.txt 2>&1
timeout 600 python tools/argmax_census.py --model l20 --fixture tests/golden/census_l20.npz --batch 1024 > $O/argmax_census_l20.txt 2>&1
echo "all done" >> $O/status.txt
# counters of the staggered-DMA GEMM variant next to the shipped kernel's (kpmc_<tag>_gemm.txt)
PCAD_ALLOW_STALE=1 PCAD_LIB=$PWD/plantcaduceus_amd/variants/libpcad_stagger.so tools/kpmc.sh gemm256q "${TAG}_gemm_stagger" > /dev/null 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/status.txt
echo "really all done" >> $O/status.txt
