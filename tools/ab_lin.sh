#!/bin/bash
# A/B of library variants (names of tools/flags_bench.py builds) on the bench launches of one kernel:
#   bash tools/ab_lin.sh lin|scan <variant> ...      ("default" is always measured first; three rounds)
what=$1; shift
tool=tools/lin_time.py; [ "$what" = scan ] && tool=tools/scan_time.py
for rep in 1 2 3; do
for v in default "$@"; do
  if [ "$v" != default ]; then export AMTGPU_LIB=amatsukaze_amd/libamt_gpu_flags_$v.so; else unset AMTGPU_LIB; fi
  echo "== $v $(python $tool 10000 8 2>/dev/null | tail -1)"
done; done
