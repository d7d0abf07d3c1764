#!/bin/bash
# round 3, GPU call 8: profile round on the final tree (bench, rocprofv3 stats from the same process, PMC passes at the benchmark's
# launch size), scan instruction-mix counters, other sizes / workloads, e2e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$ROOT/gpurun_out/r03h"; mkdir -p "$O"; cd "$ROOT"
tools/profile_round.sh r03h > "$O/profile_round.log" 2>&1
tail -5 "$O/profile_round.log"
tools/kpmc.sh scan_kernel r03h > /dev/null 2>&1
tools/kpmc.sh convx_kernel r03h_convx > /dev/null 2>&1
tools/kpmc.sh gemm256q r03h_gemm > /dev/null 2>&1
for m in l20 l24 l28; do timeout 300 python3 bench.py --model $m --cpu-seqs 0 --host-seqs 0 --no-profile 2>/dev/null | tail -1 > "$O/bench_$m.json"; done
timeout 300 python3 bench.py --dtype f32 --batch 256 --cpu-seqs 0 --host-seqs 0 --no-profile 2>/dev/null | tail -1 > "$O/bench_l32_f32.json"
for w in embed ism; do timeout 300 python3 bench.py --workload $w --cpu-seqs 0 --host-seqs 0 2>/dev/null | tail -1 > "$O/bench_$w.json"; done
timeout 600 python3 tools/e2e_5000.py > "$O/e2e_5000.json" 2> "$O/e2e_5000.err"
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value'],1), d['whole_step']['mfma_frac'], d['whole_step']['hbm_frac'])
    except Exception as e: print(f, 'failed', e)
PY
cat "$O/e2e_5000.json" | head -c 600
