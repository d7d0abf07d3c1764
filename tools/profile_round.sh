#!/bin/bash
# Run ON the GPU box (via gpurun): default bench + rocprofv3 kernel stats + three PMC passes -> gpurun_out/<tag>/
#   gpurun --timeout 1500 -- 'tools/profile_round.sh r01s'
# then here:  python profiles/pmc_summary.py gpurun_out/<tag> profiles/<tag>_pmc_traffic.json ; cp the bench JSON / stats.
set -u
TAG="${1:-prof}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
# kernel stats: the SAME process prints its own HIP-event roofline line (bench_under_rocprof.json), so the rocprofv3 average
# duration of the dominant kernel and bench.py's `roofline.achieved` come from one run (a profiled run clocks ~3-6 % lower
# than an un-profiled one: never compare across the two)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-seqs 0 --no-parity-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/kernel_stats.csv"; find "$OUT/stats" -type f ! -name "*kernel_stats.csv" -delete
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.source_hash())" > "$OUT/src_hash.txt"
# PMC passes at the BENCHMARK's launch size (--batch 1024 = two chunks of 512 windows = 524288 token-rows per launch): no scaling
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-parity-leg --no-profile --batch 1024 > "$OUT/$C.log" 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/SQ" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-parity-leg --no-profile --batch 1024 > "$OUT/SQ.log" 2>&1
# read requests by size (calibrated byte count: 32*n32 + 64*n64 + 128*n128) and the DRAM-side request counts
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/RDREQ" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-parity-leg --no-profile --batch 1024 > "$OUT/RDREQ.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/DRAM" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-parity-leg --no-profile --batch 1024 > "$OUT/DRAM.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ RDREQ DRAM; do f=$(find "$OUT/$C" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mv "$f" "$OUT/$C/p_counter_collection.csv"; find "$OUT/$C" -type f ! -name "p_counter_collection.csv" -delete; done
ls -la "$OUT" "$OUT/SQ"
