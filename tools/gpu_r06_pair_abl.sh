#!/bin/bash
# round 6: ablations of the scan's pair kernel (wrong results by design), 10 000 frames, bench launches
for v in default abl_pair_raw_sameframe abl_pair_no_sum abl_pair_no_gather abl_pair_no_flush abl_pair_no_convert abl_pair_no_eval; do
  if [ "$v" != default ]; then export AMTGPU_LIB=amatsukaze_amd/libamt_gpu_flags_$v.so; else unset AMTGPU_LIB; fi
  echo "== $v $(python tools/scan_time.py 10000 8 2>/dev/null | tail -1)"
done
