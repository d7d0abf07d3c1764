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
# round 6 additions: the secondary lines (PlantCAD2 geometries at 512 / 8 192 bp, fp32, small batches, other workloads), the
# 8 192-bp kernel tables (item 2), the parity configuration under the profiler, fuzz with random engine options, soak, smoke
tools/refresh_secondary.sh "${TAG}s" > "$O/secondary.log" 2>&1
cd /tmp; export TMPDIR=/tmp
for spec in "pc2-medium 8192 32" "pc2-medium 512 512" "pc2-large 8192 32" "pc2-large 512 512"; do
  set -- $spec
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/st_$1_$2" -- python3 "$ROOT/bench.py" --model $1 --seqlen $2 --batch $3 --steps 3 --warmup 1 --cpu-seqs 0 --host-seqs 0 --no-parity-leg > "$O/bench_$1_$2_under_rocprof.json" 2> "$O/st_$1_$2.log"
  f=$(find "$O/st_$1_$2" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats_$1_$2.csv"; rm -rf "$O/st_$1_$2"
done
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/st_f32split" -- python3 "$ROOT/bench.py" --dtype f32 --opt f32_gemm_split=1 --steps 3 --warmup 1 --cpu-seqs 0 --host-seqs 0 --no-parity-leg > "$O/bench_f32_split_under_rocprof.json" 2> "$O/st_f32split.log"
f=$(find "$O/st_f32split" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats_f32_split.csv"; rm -rf "$O/st_f32split"
cd "$ROOT"
timeout 900 python3 tools/fuzz_model.py 120 606 opts > "$O/fuzz_opts.txt" 2>&1; tail -1 "$O/fuzz_opts.txt"
timeout 600 python3 tools/fuzz_model.py 60 607 > "$O/fuzz_plain.txt" 2>&1; tail -1 "$O/fuzz_plain.txt"
timeout 600 python3 tools/fuzz_model.py 40 608 fold > "$O/fuzz_fold.txt" 2>&1; tail -1 "$O/fuzz_fold.txt"
timeout 600 python3 tools/soak.py 100 poison > "$O/soak.txt" 2>&1; tail -1 "$O/soak.txt"
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1; tail -1 "$O/smoke.txt"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_cmd.json" 2>> "$O/err.txt"
