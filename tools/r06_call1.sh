#!/bin/bash
# round 6, GPU call 1 (developer script, run through gpurun): baselines for the round's two kernel items on the round-5 kernels -
# the fp32 + f32_gemm_split configuration (bench line with per-kernel HIP-event times, SQ counters of its scans) and the
# PlantCAD2 geometries at 8 192 bp next to 512 bp under rocprofv3 --kernel-trace --stats (per-kernel microseconds per token).
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"
O=$ROOT/gpurun_out/r06a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python3 bench.py --steps 5 --warmup 2 --cpu-seqs 0 --host-seqs 0 > $O/bench_l32_bf16.json 2>> $O/err.txt
timeout 400 python3 bench.py --dtype f32 --opt f32_gemm_split=1 --steps 4 --warmup 2 --cpu-seqs 0 --host-seqs 0 > $O/bench_l32_f32_split.json 2>> $O/err.txt
cd /tmp; export TMPDIR=/tmp
for spec in "pc2-medium 8192 32" "pc2-medium 512 512" "pc2-large 8192 32" "pc2-large 512 512"; do
  set -- $spec; m=$1; L=$2; b=$3
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${m}_$L -- python3 $ROOT/bench.py --model $m --seqlen $L --batch $b --steps 3 --warmup 1 --cpu-seqs 0 --host-seqs 0 > $O/bench_${m}_${L}_under_rocprof.json 2> $O/st_${m}_$L.log
  f=$(find $O/st_${m}_$L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_${m}_$L.csv; rm -rf $O/st_${m}_$L
done
cd "$ROOT"
tools/kpmc.sh scan_kernel r06a_f32split --dtype f32 --opt f32_gemm_split=1 --batch 1024 > /dev/null 2>&1
tools/kpmc.sh scan_kernel r06a_pc2m_8192 --model pc2-medium --seqlen 8192 --batch 32 > /dev/null 2>&1
tools/kpmc.sh scan_kernel r06a_pc2m_512 --model pc2-medium --seqlen 512 --batch 512 > /dev/null 2>&1
ls -la $O
