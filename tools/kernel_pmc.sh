#!/bin/bash
# developer tool (run on the GPU box): SQ/LDS counters for one kernel of the forward:  tools/kernel_pmc.sh <kernel-name-substring>
PAT="${1:-convx_kernel}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/kpmc"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
P3="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-seqs 0 --no-profile --batch 128 --chunk-seqs 64 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p$i.csv"; rm -rf "$OUT/p$i"
done
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$OUT/p*.csv")):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("$PAT")]
    print(df.groupby("Counter_Name").Counter_Value.mean().to_string())
PY
