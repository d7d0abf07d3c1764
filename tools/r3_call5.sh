#!/bin/bash
# round 3, GPU call 5: store-pattern microbenchmark, argmax census, full GPU suite
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03e"; mkdir -p "$O"; cd "$ROOT"
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/spm tools/store_pattern_microbench.hip 2>/dev/null && /tmp/spm | tee "$O/store_patterns.txt"
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee "$O/gpu_tests.txt"
timeout 1500 python3 tools/argmax_census.py > "$O/argmax_census.txt" 2> "$O/argmax_census.err"; tail -40 "$O/argmax_census.txt"; tail -5 "$O/argmax_census.err"
