#!/bin/bash
# developer tool (run on the GPU box): SQ / LDS counters of one kernel of the forward, at the benchmark's launch size
#   tools/kpmc.sh <kernel-name-substring> <tag> [bench.py args, default: --batch 1024]   (PCAD_LIB selects the build)
# three rocprofv3 --pmc passes (8 SQ counters each); per-counter mean over the matching dispatches -> gpurun_out/kpmc_<tag>.txt
PAT="${1:-scan_kernel}"; TAG="${2:-x}"; shift 2
ARGS="${*:---batch 1024}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/kpmc_$TAG"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
P3="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-parity-leg --no-profile $ARGS > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p$i.csv"; rm -rf "$OUT/p$i"
done
python3 - <<PY | tee "$ROOT/gpurun_out/kpmc_$TAG.txt"
import pandas as pd, glob
print("kernel pattern: $PAT   bench args: $ARGS   lib: ${PCAD_LIB:-default}")
for f in sorted(glob.glob("$OUT/p*.csv")):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("$PAT")]
    # forward and reverse instantiations separately (template argument REV is the 2nd: <T, REV, ...>)
    for name, g in df.groupby(df.Kernel_Name.str.slice(0, 60)):
        print("--", name, "dispatches/counter:", g.groupby("Counter_Name").size().iloc[0])
        print(g.groupby("Counter_Name").Counter_Value.mean().to_string())
PY
rm -rf "$OUT"
