#!/bin/bash
# round 3, GPU call 9: power / clock probe, randomized-geometry fuzz and repeated-forward soak on the final tree
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03i"; mkdir -p "$O"; cd "$ROOT"
timeout 300 python3 tools/power_probe.py > "$O/power_probe.txt" 2> "$O/power_probe.err"; cat "$O/power_probe.txt"; tail -3 "$O/power_probe.err"
timeout 900 python3 tools/fuzz_model.py 60 3 > "$O/fuzz_model.txt" 2>&1; tail -4 "$O/fuzz_model.txt"
timeout 600 python3 tools/soak.py 100 > "$O/soak.txt" 2>&1; tail -2 "$O/soak.txt"
timeout 300 python3 tools/soak.py 40 poison >> "$O/soak.txt" 2>&1; tail -1 "$O/soak.txt"
