#!/bin/bash
# developer tool: build a named variant of libpcad.so with extra compiler flags (same-box A/B through PCAD_LIB)
#   tools/build_variant.sh <name> "<extra flags>" [files to rebuild with the flags, default: all]
# -> plantcaduceus_amd/variants/libpcad_<name>.so  (git-ignored; ships to the GPU box with the snapshot)
set -e
NAME="$1"; EXTRA="$2"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/plantcaduceus_amd/csrc"; OUT="$ROOT/plantcaduceus_amd/variants"; OBJ="$OUT/obj_$NAME"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
pids=()
for f in api norm gemm conv scan pack convx; do
  ( hipcc $FLAGS $EXTRA -c "$SRC/$f.hip" -o "$OBJ/$f.o" 2> "$OBJ/$f.log" || { cat "$OBJ/$f.log"; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libpcad_$NAME.so" "$OBJ"/*.o
ls -la "$OUT/libpcad_$NAME.so"
