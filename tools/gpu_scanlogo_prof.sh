#!/bin/bash
# ScanLogo per-kernel timings at 2048 and 20000 frames (and the fixed-32-frames accumulate variant when it was built):
#   bash tools/gpu_scanlogo_prof.sh <tag>      -> gpurun_out/<tag>_scanlogo.txt
TAG=${1:-r02}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_scanlogo.txt
: > $OUT
for n in 2048 20000; do
  echo "== ScanLogo $n frames (1440x1080, 256x128 rectangle)" >> $OUT
  python tools/prof_scanlogo.py --frames $n 2>&1 | grep -v amdgpu.ids >> $OUT
  if [ -f amatsukaze_amd/libamt_gpu_accfixed32.so ]; then
    echo "== same, accumulate kernel with round 1's fixed 32 frames per workgroup" >> $OUT
    AMTGPU_LIB=$PWD/amatsukaze_amd/libamt_gpu_accfixed32.so python tools/prof_scanlogo.py --frames $n 2>&1 | grep -v amdgpu.ids >> $OUT
  fi
done
cat $OUT
