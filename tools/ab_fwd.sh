#!/bin/bash
# forward / MC-dropout timing of library variants: tools/ab_fwd.sh OUTFILE lib1.so ...  ("main" = the in-tree library)
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = main ]; then unset PINN_HIP_LIB; else export PINN_HIP_LIB=$lib; fi
    echo "== $lib (rep $rep)" >> "$out"
    timeout -k 10 300 python tools/time_forward.py 2 2>&1 | grep -v amdgpu.ids >> "$out"
  done
done
