#!/bin/bash
# Run ON the GPU box (via gpurun) at the end of a round: full GPU suite + the profile round on the final tree.
#   gpurun --timeout 3600 -- 'tools/final_round.sh r04y'
# then here (the kernels' source hash must not change afterwards, or bench.py finds no matching PMC profile):
#   python profiles/pmc_summary.py gpurun_out/<tag> profiles/<tag>_pmc_traffic.json
#   python profiles/kernel_stats.py gpurun_out/<tag>/kernel_stats.csv profiles/<tag>_kernel_stats.txt "<note>"
#   cp gpurun_out/<tag>/{bench.json,bench_under_rocprof.json,gpu_tests.log,...} profiles/<tag>_*
TAG="${1:-final}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/$TAG"; mkdir -p "$O"; cd "$ROOT"
timeout 2400 python3 -m pytest tests -q -m gpu -s --durations=25 2>&1 | grep -v "^$" | tail -110 > "$O/gpu_tests.log"; tail -3 "$O/gpu_tests.log"
tools/profile_round.sh "$TAG" > "$O/profile_round.log" 2>&1; tail -2 "$O/profile_round.log"
tools/kpmc.sh scan_kernel "$TAG" > /dev/null 2>&1
tools/kpmc.sh convx_kernel "${TAG}_convx" > /dev/null 2>&1
tools/kpmc.sh gemm256q "${TAG}_gemm" > /dev/null 2>&1
for m in l20 l24 l28; do timeout 400 python3 bench.py --model $m > "$O/bench_$m.json" 2>> "$O/err.txt"; done
timeout 600 python3 tools/e2e_5000.py > "$O/e2e_5000.json" 2>> "$O/err.txt"
