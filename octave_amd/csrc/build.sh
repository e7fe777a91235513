#!/bin/bash
# Build libocta_hip.so for gfx950 (MI355X) in-tree.  Usage: build.sh [outdir]
#        build.sh diag   -> libocta_hip_diag.so: the same library with in-kernel clock stamps in the three MFMA-bound kernels
#                           (common.hpp OCTA_STAMP_*; tools/clock_stamps.py loads it through OCTA_HIP_LIB); never the shipped library
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mcode-object-version=5"
if [ "${1:-}" = "diag" ]; then
  mkdir -p "$HERE/obj/diag"
  for f in conv wgrad8; do $HIPCC $FLAGS -DOCTA_DIAG_STAMPS -c "$HERE/$f.hip" -o "$HERE/obj/diag/$f.o" & done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$HERE/../libocta_hip_diag.so" "$HERE"/obj/diag/{conv,wgrad8}.o "$HERE"/obj/{norm,pool_misc,splat_aag,loss,disc,extras,api}.o
  echo "built $HERE/../libocta_hip_diag.so"
  exit 0
fi
OUT="${1:-$HERE/..}"
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
