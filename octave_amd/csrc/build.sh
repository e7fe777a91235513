#!/bin/bash
# Build libocta_hip.so for gfx950 (MI355X) in-tree.  Usage: build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mcode-object-version=5"
mkdir -p "$HERE/obj"
pids=()
for f in conv wgrad8 norm pool_misc splat_aag loss disc extras; do
  stale=0
  for dep in "$HERE/$f.hip" "$HERE"/*.hpp "$HERE/../../include/octa_hip.h"; do
    if [ ! -f "$HERE/obj/$f.o" ] || [ "$dep" -nt "$HERE/obj/$f.o" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC $FLAGS -c -x hip "$HERE/api.cpp" -o "$HERE/obj/api.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libocta_hip.so" "$HERE"/obj/{conv,wgrad8,norm,pool_misc,splat_aag,loss,disc,extras,api}.o
echo "built $OUT/libocta_hip.so"
