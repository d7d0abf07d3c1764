#!/bin/bash
# Run ON the GPU box: the wave-specialised fused micro-kernel (MFMA waves + VALU waves in ONE dispatch) timed, then under
# rocprofv3 --pmc so both pipes' busy counters come from the same pass -> gpurun_out/coreside/
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/coreside"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
"$ROOT/tools/coreside_microbench" fused > "$OUT/fused_timing.txt" 2>&1
cat "$OUT/fused_timing.txt"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o p -- "$ROOT/tools/coreside_microbench" fused > "$OUT/pmc.log" 2>&1
f=$(find "$OUT/pmc" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee "$OUT/fused_pmc.txt"
import sys, pandas as pd
df = pd.read_csv(sys.argv[1])
df = df[df.Kernel_Name.str.contains("fused_kernel")]
p = df.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
p["kernel"] = p.Kernel_Name.str.extract(r"(fused_kernel<\d+>)")
gui = p["GRBM_GUI_ACTIVE"] / 8.0
p["valu_busy"] = p["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui
p["mfma_busy"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui
p["Mcycles"] = gui / 1e6
print(p[["Dispatch_Id", "kernel", "Mcycles", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "valu_busy", "mfma_busy"]].to_string(index=False))
PY
find "$OUT/pmc" -type f ! -name "*counter_collection.csv" -delete
