#!/bin/bash
# developer tool (GPU box): memory-side latency / stall counters of one kernel of the forward (SQ in-flight levels, TCP<->TCC request
# latencies, TLB stalls, TA stalls)      tools/kpmc_mem.sh <kernel-substring> <tag> [bench.py args]
PAT="${1:-scan_kernel}"; TAG="${2:-x}"; shift 2
ARGS="${*:---batch 1024}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/kpmcm_$TAG"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
P1="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES"
P2="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
P4="TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum"
P5="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum"
P6="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --host-seqs 0 --no-parity-leg --no-profile $ARGS > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p$i.csv"; rm -rf "$OUT/p$i"
done
python3 - <<PY | tee "$ROOT/gpurun_out/kpmcm_$TAG.txt"
import pandas as pd, glob
print("kernel pattern: $PAT   bench args: $ARGS")
for f in sorted(glob.glob("$OUT/p*.csv")):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("$PAT")]
    for name, g in df.groupby(df.Kernel_Name.str.slice(0, 64)):
        print("--", name, "dispatches/counter:", g.groupby("Counter_Name").size().iloc[0] if len(g) else 0)
        print(g.groupby("Counter_Name").Counter_Value.mean().to_string())
PY
rm -rf "$OUT"
