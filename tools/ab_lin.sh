#!/bin/bash
# A/B of library variants on the linear kernel's bench launch: bash tools/ab_lin.sh <variant> ... (names of tools/flags_bench.py builds; "default" first)
for rep in 1 2 3; do
for v in default "$@"; do
  if [ "$v" != default ]; then export AMTGPU_LIB=amatsukaze_amd/libamt_gpu_flags_$v.so; else unset AMTGPU_LIB; fi
  echo "== $v $(python tools/lin_time.py 10000 8 2>/dev/null | tail -1)"
done; done
