#!/bin/bash
# developer tool (run on the GPU box): PMC passes over tools/gemm_time.py, ours vs the vendor library kernel on the same shapes
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/gemm_pmc"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
P1="TCC_HIT TCC_MISS TCC_REQ TCC_READ TCC_TAG_STALL TCC_BUSY GRBM_GUI_ACTIVE"
P2="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_TOTAL_ACCESSES"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
P4="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/tools/gemm_time.py" > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p$i.csv"; rm -rf "$OUT/p$i"
done
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$OUT/p*.csv")):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("gemm256|Cijk_Alik_Bljk_BBS")]
    df["k"] = df.Kernel_Name.str.slice(0, 40) + " grid=" + df.Grid_Size.astype(str) + " wg=" + df.Workgroup_Size.astype(str) + " lds=" + df.LDS_Block_Size.astype(str) + " vgpr=" + df.VGPR_Count.astype(str) + "/" + df.Accum_VGPR_Count.astype(str)
    print(df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean").T.to_string())
PY
