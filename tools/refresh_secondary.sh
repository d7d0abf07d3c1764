#!/bin/bash
# developer tool (run on the GPU box): the secondary bench lines of a round on ONE tree — PlantCAD2 geometries at 512 / 8 192 bp,
# the fp32 model, small batches, the other workloads — each a plain `bench.py` invocation.   tools/refresh_secondary.sh <tag>
TAG=${1:-r04z}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { local name=$1; shift; timeout 600 python3 bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "FAILED $name" >> "$OUT/failures.txt"; }
for m in pc2-small pc2-medium pc2-large; do
  run ${m}_512 --model $m --batch 512 --steps 4 --warmup 1
done
for m in pc2-medium pc2-large; do
  run ${m}_8192_b32 --model $m --seqlen 8192 --batch 32 --steps 3 --warmup 1 --cpu-seqs 0
  run ${m}_8192_b1 --model $m --seqlen 8192 --batch 1 --steps 5 --warmup 2 --cpu-seqs 0
done
run l32_f32 --dtype f32 --cpu-seqs 0
for m in l20 l32; do
  for b in 1 8 32 128; do
    run ${m}_b$b --model $m --batch $b --steps 20 --warmup 5 --cpu-seqs 0
  done
done
run embed --workload embed --cpu-seqs 0
run ism --workload ism --cpu-seqs 0
python3 - "$OUT" <<'EOF'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):36s} {d['value']:10.1f} {d['unit']:12s} {d['ms_per_step']:9.2f} ms/step  {d['dtype']}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
EOF
