#!/bin/bash
# developer tool (run on the GPU box): the persistent GEMMs' tile walk (gemm.hip tile_walk; VERDICT r3 #1a) swept on the
# benchmark's launch size, in_proj and out_proj shapes: time (two interleaved rounds) and L2->fabric read requests per launch
# (TCC_EA0_RDREQ by request size: bytes = 32 n32 + 64 n64 + 128 n128) for every walk.  -> gpurun_out/gemm_walk.txt
#   walk = GROUP_M (1..255: m-fastest inside groups of GROUP_M m-panels) | 256 (n-fastest)
# Every walk is its own build (tools/build_variant.sh walk<w> "-DPCAD_WALK_CONST=<w>"): as a run-time kernel argument the walk
# costs the 4-wave kernel 12 % (profiles/r04_ab_runs.txt), so it is a compile-time constant of the kernel.
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/gemm_walk"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
M="${1:-524288}"; WALKS="${2:-2 4 8 16 32 256}"
V="$ROOT/plantcaduceus_amd/variants"
{
echo "# tile-walk sweep, M=$M, walks: $WALKS"
for r in 1 2; do for w in $WALKS; do
  echo "[round $r walk $w]"; PCAD_LIB="$V/libpcad_walk$w.so" timeout 300 python3 "$ROOT/tools/gemm_time.py" $M --no-vendor 2>&1 | grep "M="
done; done
for w in $WALKS; do
  PCAD_LIB="$V/libpcad_walk$w.so" timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/w$w" -o p -- python3 "$ROOT/tools/gemm_time.py" $M --no-vendor > "$OUT/w$w.log" 2>&1
  f=$(find "$OUT/w$w" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$w" <<PY
import sys, pandas as pd
df = pd.read_csv(sys.argv[1]); df = df[df.Kernel_Name.str.contains("gemm256q")]
# dispatches alternate shapes: the first 21 launches are the in_proj shape (grid identical; tell them apart by order)
piv = df.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum").sort_index()
n = len(piv) // 2
for name, part in (("in_proj  [M,1024]x[4096,1024]", piv.iloc[:n]), ("out_proj [M,2048]x[1024,2048]", piv.iloc[n:])):
    m = part.mean()
    b = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    print(f"walk {sys.argv[2]:>3} {name}: L2->fabric reads {b/1e9:.3f} GB per launch (requests {m.get('TCC_EA0_RDREQ_sum', 0):.3e})")
PY
  rm -rf "$OUT/w$w"
done
} 2>&1 | tee "$ROOT/gpurun_out/gemm_walk.txt"
