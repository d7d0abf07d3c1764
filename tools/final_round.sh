#!/bin/bash
# Run ON the GPU box (via gpurun) at the end of a round: full GPU suite + the profile round on the final tree.
#   gpurun --timeout 3000 -- 'tools/final_round.sh r03k'
TAG="${1:-final}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/$TAG"; mkdir -p "$O"; cd "$ROOT"
timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | tail -12 | tee "$O/gpu_tests.txt"
tools/profile_round.sh "$TAG" > "$O/profile_round.log" 2>&1; tail -3 "$O/profile_round.log"
tools/kpmc.sh scan_kernel "$TAG" > /dev/null 2>&1
tools/kpmc.sh convx_kernel "${TAG}_convx" > /dev/null 2>&1
